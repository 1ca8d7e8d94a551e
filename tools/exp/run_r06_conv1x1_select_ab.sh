#!/bin/bash
# A/B (same box, alternating): the 1x1 planes GEMM on the SELECTED layers of the bottleneck units (conv3 of every unit with
# filters >= 128, conv1 of the identity units with cin >= 1024 — embeddingnet_amd/backbones.py) against the gather kernels, C3.
out=gpurun_out/r06_exp_conv1x1_select.txt
: > $out
python -m pytest tests/test_conv1x1_planes_gpu.py -q 2>&1 | tail -8 >> $out
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== C3 EMBNET_CONV_1X1_PLANES=$v rep $rep" >> $out
    EMBNET_CONV_1X1_PLANES=$v python bench.py --config c3 --steps 12 --warmup 4 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep -E "enqueue loop|conv1x1_planes_kernel|conv_fwd_h_kernel|affine_act_planes|traced kernels|\"metric\"" | cut -c1-260 >> $out
  done
done
