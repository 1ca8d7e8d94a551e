# BatchNorm-backward sums from the conv data gradient (EMBNET_FUSE_BN_SUMS): tests, then C3 / C2 / C5 with and without
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps ${STEPS:-40} --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_backbone_gpu.py -q -m gpu -x 2>&1 | tail -4
for i in 1 2; do
  STEPS=12 one BCFG=c3 EMBNET_FUSE_BN_SUMS=0
  STEPS=12 one BCFG=c3 EMBNET_FUSE_BN_SUMS=1
  one BCFG=c2 EMBNET_FUSE_BN_SUMS=0
  one BCFG=c2 EMBNET_FUSE_BN_SUMS=1
  one BCFG=c5 EMBNET_FUSE_BN_SUMS=0
  one BCFG=c5 EMBNET_FUSE_BN_SUMS=1
done
