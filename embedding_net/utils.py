"""Re-export of embeddingnet_amd.utils under the reference's package name."""
from embeddingnet_amd.utils import *  # noqa: F401,F403
