#!/bin/bash
# The stem's forward on 256 x 64 tiles (EMBNET_FWD_256, three-product form only): per-launch and C2 / C3 step, same box, alternating.
out=gpurun_out/r06_exp_fwd256.txt
: > $out
for v in 0 1; do
  echo "== stem launch, EMBNET_FWD_256=$v" >> $out
  EMBNET_FWD_256=$v python tools/exp/step_launch_list.py resnet18 128 224 conv_fwd_h 2>&1 | head -3 | cut -c1-200 >> $out
done
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== C2 EMBNET_FWD_256=$v rep $rep" >> $out
    EMBNET_FWD_256=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batch-hard --sustain-seconds 0 2>&1 | grep -E "conv_fwd_h_kernel|traced kernels|\"metric\"" | cut -c1-230 >> $out
  done
done
for v in 0 1; do
  echo "== C3 EMBNET_FWD_256=$v" >> $out
  EMBNET_FWD_256=$v python bench.py --config c3 --steps 12 --warmup 4 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep -E "traced kernels|\"metric\"" | cut -c1-230 >> $out
done
python -m pytest tests/test_backbone_gpu.py tests/test_full_size_gpu.py tests/test_conv_ranges_gpu.py -q -x 2>&1 | tail -4 >> $out
