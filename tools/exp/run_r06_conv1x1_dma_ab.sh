#!/bin/bash
# The fp32-by-DMA 1x1 kernel (conv1x1_a32_kernel): tests, back-to-back table, then C3 alternating EMBNET_CONV_1X1_DMA_MODE = 0 / 1 / 2
# on one box (0: gather kernels; 1: forward + data gradients without BatchNorm sums; 2: those too).
out=gpurun_out/r06_exp_conv1x1_dma_step.txt
: > $out
python -m pytest tests/test_conv1x1_dma_gpu.py -q 2>&1 | tail -12 >> $out
python tools/exp/conv1x1_planes_bench.py > gpurun_out/r06_exp_conv1x1_dma.txt 2>&1
for rep in 1 2; do
  for v in 0 1 2; do
    echo "== C3 EMBNET_CONV_1X1_DMA_MODE=$v rep $rep" >> $out
    EMBNET_CONV_1X1_DMA_MODE=$v python bench.py --config c3 --steps 12 --warmup 4 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep -E "enqueue loop|conv1x1_a32|conv_fwd_h_kernel|conv_dgrad_h_kernel|bn_bwd_reduce|traced kernels|\"metric\"" | cut -c1-230 >> $out
  done
done
