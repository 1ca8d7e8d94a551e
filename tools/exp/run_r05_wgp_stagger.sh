#!/bin/bash
# Stagger variant of the planes weight gradient (EMBNET_WGP_KNOBS=4): correctness, back-to-back and in-step C2 A/B.
set -u
mkdir -p gpurun_out
O=gpurun_out/r05_exp_wgp_stagger.txt
: > $O
echo "== correctness (EMBNET_WGP_KNOBS=4)" >> $O
EMBNET_WGP_KNOBS=4 timeout 600 python -m pytest tests/test_wgrad_planes_gpu.py -q -x 2>&1 | tail -3 >> $O
for k in 0 4 0 4; do
  echo "== back-to-back knobs=$k" >> $O
  EMBNET_WGP_KNOBS=$k timeout 300 python tools/exp/wgrad_planes_bench.py 2>&1 | tail -9 >> $O
done
for r in 1 2 3; do
  for k in 0 4; do
    echo "== c2 in-step knobs=$k round=$r" >> $O
    BCFG=c2 EMBNET_WGP_KNOBS=$k timeout 300 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O
  done
done
cat $O
