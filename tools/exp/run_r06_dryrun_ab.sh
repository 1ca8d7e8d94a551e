#!/bin/bash
# A/B: the BatchNorm backward's planes scaled from a bound (EMBNET_BN_BWD_BOUND=1) against round 5's dry run of the apply pass (=0).
out=gpurun_out/r06_exp_bn_bwd_bound.txt
: > $out
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== C2 EMBNET_BN_BWD_BOUND=$v rep $rep" >> $out
    EMBNET_BN_BWD_BOUND=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batch-hard --sustain-seconds 0 2>&1 | grep -E "enqueue loop|bn_bwd_apply4|bn_bwd_reduce4|traced kernels|\"metric\"" | cut -c1-230 >> $out
  done
done
for v in 0 1; do
  echo "== C3 EMBNET_BN_BWD_BOUND=$v" >> $out
  EMBNET_BN_BWD_BOUND=$v python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep -E "enqueue loop|bn_bwd_apply4|bn_bwd_reduce4|traced kernels|\"metric\"" | cut -c1-230 >> $out
done
