# in-step timing of the small-backbone configs (and C2 as the regression check): `bash tools/exp/run_r04_small_nets.sh`
one() { echo -n "$* : "; env "$@" python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
for i in 1 2; do
  one X=1 BCFG=c1s; one X=1 BCFG=c1; one X=1 BCFG=c2
done
one EMBNET_CONV_TILE=1 BCFG=c1s
one EMBNET_CONV_TILE=3 BCFG=c1
one EMBNET_CONV_TILE=1 BCFG=c1
