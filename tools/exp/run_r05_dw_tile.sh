#!/bin/bash
# LDS-tile depthwise kernel (csrc/dwconv_tile.hip) against the row kernels: back to back per layer, then C5 in the step.
set -u
mkdir -p gpurun_out
O=gpurun_out/r05_exp_dw_tile.txt
: > $O
for t in 0 1 0 1; do
  echo "== back-to-back EMBNET_DW_TILE=$t" >> $O
  EMBNET_DW_TILE=$t timeout 300 python tools/exp/dw_tile_bench.py 2>&1 | grep '^{' >> $O
done
echo "== back-to-back EMBNET_DW_TILE=1 EMBNET_DW_TILE_WGRAD=0" >> $O
EMBNET_DW_TILE_WGRAD=0 timeout 300 python tools/exp/dw_tile_bench.py 2>&1 | grep '^{' >> $O
for r in 1 2 3; do
  echo "== c5 in-step EMBNET_DW_TILE=1 EMBNET_DW_TILE_WGRAD=0 round=$r" >> $O
  BCFG=c5 EMBNET_DW_TILE_WGRAD=0 timeout 300 python bench.py --steps 30 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O
  for t in 0 1; do
    echo "== c5 in-step EMBNET_DW_TILE=$t round=$r" >> $O
    BCFG=c5 EMBNET_DW_TILE=$t timeout 300 python bench.py --steps 30 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O
  done
done
cat $O
