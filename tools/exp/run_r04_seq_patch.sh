# simple2's 3x3 stride-1 convs (conv2, conv5) on the patch kernel (forward only): in-step A/B on C1
one() { echo -n "$* : "; env "$@" timeout 600 python bench.py --steps 60 --no-cpu-baseline --sustain-seconds 0 2>gpurun_out/seq_patch_$2.err | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
for i in 1 2; do
  one BCFG=c1 EMBNET_SEQ_PATCH=0
  one BCFG=c1 EMBNET_SEQ_PATCH=1
done
grep "conv_\|affine" gpurun_out/seq_patch_EMBNET_SEQ_PATCH=1.err | cut -c1-200
