python tools/exp/ab_conv.py build_variants/occ0.so build_variants/occ1.so > gpurun_out/ab_occ.txt 2>&1
cat gpurun_out/ab_occ.txt
python -m pytest tests/test_step_parity_gpu.py -m gpu -q --durations=6 -k "curve" > gpurun_out/pytest_r02d.txt 2>&1; tail -12 gpurun_out/pytest_r02d.txt | cut -c1-250
