# stride-1 depthwise data gradient emits the previous BatchNorm's backward sums (EMBNET_DW_BN_SUMS): tests, then C5 with / without
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_mbconv_siamese_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_step_parity_gpu.py tests/test_round3_gpu.py tests/test_backbone_gpu.py -q -m gpu -x -k "efficientnet" 2>&1 | tail -2
for i in 1 2 3; do
  one BCFG=c5 EMBNET_DW_BN_SUMS=0
  one BCFG=c5 EMBNET_DW_BN_SUMS=1
done
