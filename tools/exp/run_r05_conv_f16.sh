#!/bin/bash
# Three products on the gather convs (EMBNET_CONV_F16=1, default) against the six-term split (=0), alternating, C2 and C3;
# first the new parity tests and the ResNet whole-net / step tests that now run on the three-product kernels.
set -u
mkdir -p gpurun_out
O=gpurun_out/r05_exp_conv_f16.txt
: > $O
timeout 900 python -m pytest tests/test_conv_ranges_gpu.py -x -q -s 2>&1 | tail -40 > gpurun_out/r05_conv_f16_tests.txt
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_step_context_gpu.py tests/test_conv_patch_gpu.py -x -q -m gpu -k "resnet or Resnet or step or c3 or siamese" 2>&1 | tail -15 >> gpurun_out/r05_conv_f16_tests.txt
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print(d['value'], d['ms_per_step'], 'loss', c.get('loss_first_timed'), '->', c.get('loss_last_timed'), '|', r['kernel'][:64], r['avg_us'], r['frac'])"; }
for r in 1 2 3; do for f in 0 1; do
  echo "== c2 EMBNET_CONV_F16=$f round=$r" >> $O
  BCFG=c2 EMBNET_CONV_F16=$f timeout 300 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 2>gpurun_out/r05_conv_f16_c2_$f.err | line >> $O
done; done
for f in 0 1; do
  echo "== c3 EMBNET_CONV_F16=$f" >> $O
  BCFG=c3 EMBNET_CONV_F16=$f timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 2>gpurun_out/r05_conv_f16_c3_$f.err | line >> $O
done
cat gpurun_out/r05_conv_f16_tests.txt; cat $O; grep -h "conv_\|bn_bwd_apply4\|range" gpurun_out/r05_conv_f16_c2_1.err | head -30
