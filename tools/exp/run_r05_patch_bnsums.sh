#!/bin/bash
# BatchNorm-backward sums from the patch kernel's data-gradient epilogue: parity tests, then C2 / C3 in-step A/B (python flag).
set -u
mkdir -p gpurun_out
O=gpurun_out/r05_exp_patch_bnsums.txt
: > $O
timeout 900 python -m pytest tests/test_conv_patch_gpu.py -q -x 2>&1 | tail -4 >> $O
for r in 1 2 3; do
  for t in 0 1; do
    echo "== c2 in-step EMBNET_PATCH_BN_SUMS=$t round=$r" >> $O
    BCFG=c2 EMBNET_PATCH_BN_SUMS=$t timeout 300 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O
  done
done
for t in 0 1; do
  echo "== c3 in-step EMBNET_PATCH_BN_SUMS=$t" >> $O
  BCFG=c3 EMBNET_PATCH_BN_SUMS=$t timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O
done
echo "== c2 kernel table (flag on)" >> $O
BCFG=c2 timeout 300 python bench.py --steps 20 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep -E "x +[0-9]+/step|traced kernels" | head -14 >> $O
cat $O
