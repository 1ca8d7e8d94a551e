#!/bin/bash
# C2 with the three-product build: eager step (the probe's choice at N = 1) against the captured step (--force-graph), alternating.
set -u
mkdir -p gpurun_out
O=gpurun_out/r05_exp_graph_ab.txt
: > $O
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], d['ms_per_step'], 'mode', c.get('step_mode'), 'host', c.get('host_enqueue_ms_per_step'), 'loss', c.get('loss_last_timed'))"; }
for r in 1 2 3; do
  echo "== c2 auto round=$r" >> $O
  timeout 300 python bench.py --steps 50 --no-cpu-baseline --sustain-seconds 0 2>gpurun_out/r05_graph_ab_auto.err | line >> $O
  grep -h "step mode\|enqueue loop" gpurun_out/r05_graph_ab_auto.err >> $O
  echo "== c2 --force-graph round=$r" >> $O
  timeout 300 python bench.py --steps 50 --no-cpu-baseline --sustain-seconds 0 --force-graph 2>gpurun_out/r05_graph_ab_graph.err | line >> $O
  grep -h "step mode\|enqueue loop" gpurun_out/r05_graph_ab_graph.err >> $O
done
cat $O
