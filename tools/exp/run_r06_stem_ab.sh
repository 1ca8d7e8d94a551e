#!/bin/bash
# The stem's forward on its own kernel (csrc/conv_stem.hip; EMBNET_STEM_CONV=0: the gather kernel): tests, then C2 / C3 alternating, same box.
out=gpurun_out/r06_exp_stem.txt
: > $out
python -m pytest tests/test_conv_stem_gpu.py tests/test_backbone_gpu.py tests/test_full_size_gpu.py -q 2>&1 | tail -5 >> $out
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== C2 EMBNET_STEM_CONV=$v rep $rep" >> $out
    EMBNET_STEM_CONV=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batch-hard --sustain-seconds 0 2>&1 | grep -E "conv_stem_kernel|conv_fwd_h_kernel|traced kernels|\"metric\"" | cut -c1-230 >> $out
  done
done
for rep in 1 2; do
  for v in 0 1; do
    echo "== C3 EMBNET_STEM_CONV=$v rep $rep" >> $out
    EMBNET_STEM_CONV=$v python bench.py --config c3 --steps 12 --warmup 4 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep -E "conv_stem_kernel|traced kernels|\"metric\"" | cut -c1-230 >> $out
  done
done
