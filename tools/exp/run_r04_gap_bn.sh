# pooled (squeeze-excite) gradient folded into the BatchNorm backward passes (EMBNET_FUSE_GAP_BN): test, then C5 with / without
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 600 python -m pytest tests/test_round4_gpu.py tests/test_mbconv_siamese_gpu.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2 3; do
  one BCFG=c5 EMBNET_FUSE_GAP_BN=0
  one BCFG=c5 EMBNET_FUSE_GAP_BN=1
done
