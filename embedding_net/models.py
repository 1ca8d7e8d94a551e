"""Re-export of embeddingnet_amd.models under the reference's package name."""
from embeddingnet_amd.models import *  # noqa: F401,F403
