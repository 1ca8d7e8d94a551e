"""Re-export of embeddingnet_amd.backbones under the reference's package name."""
from embeddingnet_amd.backbones import *  # noqa: F401,F403
