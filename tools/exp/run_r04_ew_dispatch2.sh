one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps ${STEPS:-40} --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
OLD=$PWD/build_variants/ew_old.so
for i in 1 2 3; do
  for cfg in c5 c2 c1; do
    one BCFG=$cfg EMBNET_LIB=$OLD
    one BCFG=$cfg
  done
  STEPS=12 one BCFG=c3 EMBNET_LIB=$OLD
  STEPS=12 one BCFG=c3
done
