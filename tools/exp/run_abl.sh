python tools/exp/ab_conv.py build_variants/wg_x0.so build_variants/abl1.so build_variants/abl2.so build_variants/abl3.so --rounds 3 > gpurun_out/ab_ablate.txt 2>&1
python tools/exp/ab_conv.py build_variants/wg_x0.so build_variants/wg_x1s0.so build_variants/wg_x1s1.so build_variants/wg_x1s3.so --only wgrad --rounds 3 > gpurun_out/ab_wgrad_xcd.txt 2>&1
cat gpurun_out/ab_ablate.txt gpurun_out/ab_wgrad_xcd.txt
