# BatchNorm-backward sums from the PATCH kernel's data gradient: tests, then C2 / C3 — new build with the fusion, new build without
# (EMBNET_PATCH_BN_SUMS=0: is the kernel itself slower for carrying the extra epilogue?), and the previous build
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps ${STEPS:-40} --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
echo skip tests
OLD=$PWD/build_variants/pbn_old.so
for i in 1 2 3; do
  one BCFG=c2 EMBNET_LIB=$OLD EMBNET_LIB_LAX=1 EMBNET_PATCH_BN_SUMS=0
  one BCFG=c2 EMBNET_PATCH_BN_SUMS=0
  one BCFG=c2 EMBNET_PATCH_BN_SUMS=1
done
for i in 1 2; do
  STEPS=12 one BCFG=c3 EMBNET_LIB=$OLD EMBNET_LIB_LAX=1 EMBNET_PATCH_BN_SUMS=0
  STEPS=12 one BCFG=c3 EMBNET_PATCH_BN_SUMS=1
done
