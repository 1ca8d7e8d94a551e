#!/bin/bash
# The small backbones on three products (ABI 22): tests, then C1 / C1s alternating with EMBNET_CONV_F16=0 (six terms) on one box.
out=gpurun_out/r06_exp_small_backbones.txt
: > $out
python -m pytest tests/test_small_backbone_ranges_gpu.py tests/test_backbone_gpu.py tests/test_conv_ranges_gpu.py tests/test_activation_range_gpu.py -x -q 2>&1 | tail -15 >> $out
for rep in 1 2; do
  for cfg in c1s c1; do
    for v in 0 1; do
      echo "== $cfg EMBNET_CONV_F16=$v rep $rep" >> $out
      EMBNET_CONV_F16=$v python bench.py --config $cfg --steps 200 --warmup 30 --no-cpu-baseline --sustain-seconds 0 --force-graph 2>&1 | grep -E "enqueue loop|step mode|traced kernels|\"metric\"" | cut -c1-250 >> $out
    done
  done
done
