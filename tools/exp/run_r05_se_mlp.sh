#!/bin/bash
# Fused squeeze-and-excite gate (csrc/se_mlp.hip) against the composed dense / activation launches: C5 in the step, kernel table.
set -u
mkdir -p gpurun_out
O=gpurun_out/r05_exp_se_mlp.txt
: > $O
for r in 1 2 3; do
  for t in 0 1; do
    echo "== c5 in-step EMBNET_SE_MLP=$t round=$r" >> $O
    BCFG=c5 EMBNET_SE_MLP=$t timeout 300 python bench.py --steps 30 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'host enqueue', d['config'].get('host_enqueue_ms_per_step'))" >> $O
  done
done
for t in 0 1; do
  echo "== c5 kernel trace EMBNET_SE_MLP=$t (se / dense / act / colsum rows)" >> $O
  BCFG=c5 EMBNET_SE_MLP=$t EMBNET_BENCH_ROWS=60 timeout 300 python bench.py --steps 10 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep -E "semlp|dense_|act_fwd|act_bwd|colsum|step mode|enqueue loop|traced kernels" >> $O
done
cat $O
