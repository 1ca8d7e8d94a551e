"""Re-export of embeddingnet_amd.parallel under the reference's package name."""
from embeddingnet_amd.parallel import *  # noqa: F401,F403
