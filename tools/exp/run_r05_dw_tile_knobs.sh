#!/bin/bash
# LDS-tile depthwise kernel: workgroup count / units per workgroup / band height sweeps (back to back).
set -u
mkdir -p gpurun_out
O=gpurun_out/r05_exp_dw_tile_knobs.txt
: > $O
run() { echo "== $*" >> $O; env "$@" timeout 300 python tools/exp/dw_tile_bench.py --iters 30 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['h'], d['c'], d['k'], 'fwd', d['fwd_us'], d['fwd_GBs'], 'rows', d['fwd_rows'], 'dgrad', d['dgrad_us'], d['dgrad_GBs'])" >> $O; }
run EMBNET_DW_TILE=1
run EMBNET_DW_TILE=1 EMBNET_DW_TILE_MIN_UNITS=1
run EMBNET_DW_TILE=1 EMBNET_DW_TILE_MIN_UNITS=2
run EMBNET_DW_TILE=1 EMBNET_DW_TILE_MIN_UNITS=8
run EMBNET_DW_TILE=1 EMBNET_DW_TILE_MIN_UNITS=1 EMBNET_DW_TILE_BLOCKS=768
run EMBNET_DW_TILE=1 EMBNET_DW_TILE_MIN_UNITS=1 EMBNET_DW_TILE_BLOCKS=1024
run EMBNET_DW_TILE=1 EMBNET_DW_TILE_MIN_UNITS=1 EMBNET_DW_TILE_BLOCKS=2304
run EMBNET_DW_TILE=1 EMBNET_DW_TILE_MIN_UNITS=1 EMBNET_DW_TILE_BLOCKS=4096
run EMBNET_DW_TILE=1 EMBNET_DW_TILE_BAND=14
run EMBNET_DW_TILE=1 EMBNET_DW_TILE_BAND=7
run EMBNET_DW_TILE=1 EMBNET_DW_TILE_BAND=14 EMBNET_DW_TILE_MIN_UNITS=2
cat $O
