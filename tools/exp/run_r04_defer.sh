# BatchNorm apply deferred into the consuming convs' loaders (EMBNET_DEFER_BN=1) after the tf kernels got their occupancy bound
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps ${STEPS:-40} --no-cpu-baseline --sustain-seconds 0 2>gpurun_out/defer_$2_$3.err | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
for i in 1 2; do
  STEPS=12 one BCFG=c3 EMBNET_DEFER_BN=0
  STEPS=12 one BCFG=c3 EMBNET_DEFER_BN=2
  one BCFG=c2 EMBNET_DEFER_BN=0
  one BCFG=c2 EMBNET_DEFER_BN=2
done
grep "conv_\|affine\|bn_" gpurun_out/defer_EMBNET_DEFER_BN=2_.err | cut -c1-190
