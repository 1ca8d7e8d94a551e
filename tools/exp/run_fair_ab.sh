python tools/exp/ab_conv.py build_variants/fair0.so build_variants/fair1.so > gpurun_out/ab_fair.txt 2>&1
EMBNET_LIB=build_variants/stamps_fair1.so python tools/exp/conv_timeline.py --warm 200 > gpurun_out/timeline_r02e_fair1.txt 2>&1
cat gpurun_out/ab_fair.txt
