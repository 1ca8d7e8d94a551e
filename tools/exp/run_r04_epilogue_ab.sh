# Epilogue operand loads hoisted out of the staging loop (conv fwd / dgrad, pairwise, dense): parity tests, then the same
# steps with the previous build (build_variants/epi_old.so = HEAD~ library) and the new one, alternating on one box.
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps ${STEPS:-40} --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 1500 python -m pytest tests/test_backbone_gpu.py tests/test_conv_patch_gpu.py tests/test_edge_cases_gpu.py tests/test_loss_path_gpu.py tests/test_round4_gpu.py tests/test_full_size_gpu.py -q -m gpu -x 2>&1 | tail -4
OLD=$PWD/build_variants/epi_old.so
for i in 1 2; do
  for cfg in c1 c1s c2 c5; do
    one BCFG=$cfg EMBNET_LIB=$OLD
    one BCFG=$cfg
  done
  STEPS=12 one BCFG=c3 EMBNET_LIB=$OLD
  STEPS=12 one BCFG=c3
done
for lib in $OLD ""; do
  echo "pairwise sweep lib=${lib:-new}"
  EMBNET_LIB=$lib python tools/kernel_bench.py gemm --n 1024 4096 16384 --e 256 512 4096 2>/dev/null | grep pairwise
done
