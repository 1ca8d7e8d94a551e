#!/bin/bash
# The planes' formats in the step: two fp16 pieces + scale / three products (default) against three bf16 pieces / six products
# (EMBNET_PLANES_F16=0), alternating, C2 and C3 (and C1 / C1s / C5 once each: their 3x3 stride-1 convs, if any, follow the format).
set -u
mkdir -p gpurun_out
O=gpurun_out/r05_exp_planes_f16.txt
: > $O
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print(d['value'], d['ms_per_step'], 'loss', c.get('loss_first_timed'), '->', c.get('loss_last_timed'), '|', r['kernel'][:64], r['avg_us'], r['frac'])"; }
for r in 1 2 3; do for f in 0 1; do
  echo "== c2 EMBNET_PLANES_F16=$f round=$r" >> $O
  BCFG=c2 EMBNET_PLANES_F16=$f timeout 300 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | line >> $O
done; done
for f in 0 1; do
  echo "== c3 EMBNET_PLANES_F16=$f" >> $O
  BCFG=c3 EMBNET_PLANES_F16=$f timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | line >> $O
done
for c in c1 c1s c5; do for f in 0 1; do
  echo "== $c EMBNET_PLANES_F16=$f" >> $O
  BCFG=$c EMBNET_PLANES_F16=$f timeout 300 python bench.py --steps 20 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | line >> $O
done; done
cat $O
