"""Re-export of embeddingnet_amd.losses_and_accuracies under the reference's package name."""
from embeddingnet_amd.losses_and_accuracies import *  # noqa: F401,F403
