"""Re-export of embeddingnet_amd.datagenerators under the reference's package name."""
from embeddingnet_amd.datagenerators import *  # noqa: F401,F403
