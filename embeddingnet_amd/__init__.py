"""embeddingnet_amd — MI355X-native metric-learning training hot path.

Mirrors the Python surface of RocketFlash/EmbeddingNet's embedding_net package
(losses_and_accuracies, backbones, models, datagenerators, utils) on top of
libembnet_hip.so (hand-written gfx950 HIP kernels, C ABI in include/embnet.h).
"""
__version__ = "0.1.0"
