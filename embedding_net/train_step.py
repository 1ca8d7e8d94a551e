"""Re-export of embeddingnet_amd.train_step under the reference's package name."""
from embeddingnet_amd.train_step import *  # noqa: F401,F403
