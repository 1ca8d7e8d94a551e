# element-wise BatchNorm kernels with their run-time switches resolved by one dispatch in front of the loop: A/B in-step
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps ${STEPS:-40} --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_round4_gpu.py tests/test_mbconv_siamese_gpu.py -q -m gpu -x 2>&1 | tail -2
OLD=$PWD/build_variants/ew_old.so
echo old; EMBNET_LIB=$OLD python tools/exp/time_bn.py 2>/dev/null | cut -c1-230
echo new; python tools/exp/time_bn.py 2>/dev/null | cut -c1-230
for i in 1 2; do
  for cfg in c2 c5 c1; do
    one BCFG=$cfg EMBNET_LIB=$OLD
    one BCFG=$cfg
  done
  STEPS=12 one BCFG=c3 EMBNET_LIB=$OLD
  STEPS=12 one BCFG=c3
done
