one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
OLD=$PWD/build_variants/c2_ref.so
for i in 1 2 3; do
  one BCFG=c2 EMBNET_LIB=$OLD EMBNET_LIB_LAX=1 EMBNET_FUSE_BN_SUMS=0
  one BCFG=c2
done
