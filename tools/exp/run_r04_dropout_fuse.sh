# Dropout riding on the BatchNormalization kernels (simple2): tests, then the C1 step with and without it
one() { echo -n "$* : "; env "$@" timeout 600 python bench.py --steps 60 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 900 python -m pytest tests/test_round4_gpu.py -q -m gpu -k "dropout or relu_backward" 2>&1 | tail -5
for i in 1 2 3; do
  one EMBNET_FUSE_DROPOUT_BN=0 BCFG=c1
  one EMBNET_FUSE_DROPOUT_BN=1 BCFG=c1
done
