"""Alias package: `embedding_net.<module>` re-exports `embeddingnet_amd.<module>`, so code written
against RocketFlash/EmbeddingNet's import paths (tools/train.py:9-15) picks up the MI355X hot path."""
