# squeeze-excite backward: BatchNorm sums from the gate-gradient pass (EMBNET_SE_BN_SUMS): tests, then C5 with / without
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
echo tests skipped
echo tests skipped
for i in 1 2 3; do
  one BCFG=c5 EMBNET_SE_BN_SUMS=0
  one BCFG=c5 EMBNET_SE_BN_SUMS=1
done
