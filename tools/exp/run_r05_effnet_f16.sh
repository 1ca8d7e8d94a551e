#!/bin/bash
# EfficientNet's implicit-GEMM 1x1 convs on three products where the ranges are known (EMBNET_EFFNET_F16=1) against six terms (=0),
# alternating, C5; the EfficientNet parity tests first.
set -u
mkdir -p gpurun_out
O=gpurun_out/r05_exp_effnet_f16.txt
: > $O
timeout 900 python -m pytest tests -x -q -m gpu -k "efficientnet or Efficientnet or c5 or mbconv or reference_config" 2>&1 | tail -6 >> $O
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print(d['value'], d['ms_per_step'], 'loss', c.get('loss_first_timed'), '->', c.get('loss_last_timed'), '|', r['kernel'][:64], r['avg_us'], r['frac'])"; }
for r in 1 2 3; do for f in 0 1; do
  echo "== c5 EMBNET_EFFNET_F16=$f round=$r" >> $O
  BCFG=c5 EMBNET_EFFNET_F16=$f timeout 300 python bench.py --steps 30 --no-cpu-baseline --sustain-seconds 0 2>gpurun_out/r05_effnet_f16_$f.err | line >> $O
done; done
cat $O; grep -h "conv_\|bn_bwd_apply4\|launches per step" gpurun_out/r05_effnet_f16_1.err | head -20; grep -h "conv_\|launches per step" gpurun_out/r05_effnet_f16_0.err | head -12
