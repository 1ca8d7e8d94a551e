python tools/exp/ab_conv.py build_variants/prio0.so build_variants/prio1.so > gpurun_out/ab_prio.txt 2>&1
EMBNET_LIB=build_variants/stamps_prio1.so python tools/exp/conv_timeline.py --warm 200 > gpurun_out/timeline_r02d_prio1.txt 2>&1
cat gpurun_out/ab_prio.txt
