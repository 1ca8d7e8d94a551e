#!/bin/bash
# The weight gradients on a side stream (EMBNET_OVERLAP_WGRAD=1: they overlap the HBM-bound BatchNorm-backward passes of the layer in front), same box, alternating.
out=gpurun_out/r06_exp_overlap_wgrad.txt
: > $out
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== C2 EMBNET_OVERLAP_WGRAD=$v rep $rep" >> $out
    EMBNET_OVERLAP_WGRAD=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batch-hard --no-kernel-timer --sustain-seconds 0 2>&1 | grep -E "enqueue loop|\"metric\"" | cut -c1-230 >> $out
  done
done
for rep in 1 2; do
  for v in 0 1; do
    echo "== C3 EMBNET_OVERLAP_WGRAD=$v rep $rep" >> $out
    EMBNET_OVERLAP_WGRAD=$v python bench.py --config c3 --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timer --sustain-seconds 0 2>&1 | grep -E "enqueue loop|\"metric\"" | cut -c1-230 >> $out
  done
done
