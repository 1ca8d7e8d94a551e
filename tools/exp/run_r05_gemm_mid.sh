# distance GEMM at 512 <= N < 4096: tile / K-split rule sweep, then the full sweep with the defaults
sweep() { echo "== $*"; env "$@" timeout 300 python tools/kernel_bench.py gemm --n 512 1024 2048 3072 --e 256 512 4096 --iters 30 2>/dev/null | grep pairwise; }
timeout 600 python -m pytest tests/test_loss_path_gpu.py tests/test_knn_gpu.py tests/test_oracle_golden.py -q -x 2>&1 | tail -3
sweep EMBNET_PAIRWISE_GL_MIN_TILES=1 EMBNET_PAIRWISE_SPLIT_TARGET=1      # round 4: 128x128 tiles from N = 1024... (all sizes here)
sweep EMBNET_PAIRWISE_SPLIT_TARGET=1                                      # 64x64 tiles, no split
sweep EMBNET_PAIRWISE_SPLIT_TARGET=512
sweep EMBNET_PAIRWISE_SPLIT_TARGET=1024
sweep EMBNET_PAIRWISE_SPLIT_TARGET=1024 EMBNET_PAIRWISE_MIN_KT=2
sweep EMBNET_PAIRWISE_SPLIT_TARGET=768 EMBNET_PAIRWISE_GL_MIN_TILES=256
timeout 600 python tools/kernel_bench.py gemm --json gpurun_out/r05_gemm_sweep.json 2>/dev/null | tail -24
