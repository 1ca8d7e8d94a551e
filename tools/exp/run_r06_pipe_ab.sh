#!/bin/bash
# A/B of the pipelined fragment reads of the patch kernel (EMBNET_PATCH_PIPE): same box, alternating, C2 and C3.
out=gpurun_out/r06_exp_patch_pipe.txt
: > $out
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== C2 EMBNET_PATCH_PIPE=$v rep $rep" >> $out
    EMBNET_PATCH_PIPE=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep -E "step mode|enqueue loop|conv_patch_kernel|wgrad_planes|traced kernels|\"metric\"" | cut -c1-230 >> $out
  done
done
for v in 0 1; do
  echo "== C3 EMBNET_PATCH_PIPE=$v" >> $out
  EMBNET_PATCH_PIPE=$v python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep -E "enqueue loop|conv_patch_kernel|traced kernels|\"metric\"" | cut -c1-230 >> $out
done
