# tile choice for short reductions (EMBNET_CONV_SHORTK): C5 and C3 A/B
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 30 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
for i in 1 2; do
  one BCFG=c5 EMBNET_CONV_SHORTK=0
  one BCFG=c5 EMBNET_CONV_SHORTK=1
  one BCFG=c5 EMBNET_CONV_SHORTK=3
  one BCFG=c5 EMBNET_CONV_SHORTK=3 EMBNET_CONV_SHORTK_MAX=16
done
one BCFG=c3 EMBNET_CONV_SHORTK=0
one BCFG=c3 EMBNET_CONV_SHORTK=1
one BCFG=c3 EMBNET_CONV_SHORTK=3
