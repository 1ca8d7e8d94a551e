#!/bin/bash
# Rows of statistics partials per depthwise row-kernel launch (EMBNET_DW_STATS_ROWS): stride-2 data gradients back to back, C5 in the step.
set -u
mkdir -p gpurun_out
O=gpurun_out/r05_exp_dw_stats_rows.txt
: > $O
for t in 0 1024 2048 4096 8192; do
  echo "== back-to-back EMBNET_DW_STATS_ROWS=$t (0: one row per 256 threads, the old form)" >> $O
  EMBNET_DW_STATS_ROWS=$t timeout 300 python tools/exp/dw_s2_dgrad_bench.py 2>&1 | grep '^{' >> $O
done
for r in 1 2; do
  for t in 0 2048 4096; do
    echo "== c5 in-step EMBNET_DW_STATS_ROWS=$t round=$r" >> $O
    BCFG=c5 EMBNET_DW_STATS_ROWS=$t timeout 300 python bench.py --steps 30 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O
  done
done
cat $O
