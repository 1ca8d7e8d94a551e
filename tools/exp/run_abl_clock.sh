for v in stamps stamps_abl1 stamps_abl3; do
  for sh in 128,56,56,64,3,64,1,1 128,14,14,256,3,256,1,1; do
    EMBNET_LIB=build_variants/$v.so python tools/exp/conv_timeline.py --warm 300 --shape $sh --only fwd 2>&1 | grep kernel | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$v', d['kernel'], d['tflops'],'TF clk',d['clock_mhz'],'busy',d['cu_pipe_busy'],'loop/kt',d['loop_cyc_per_ktile'],'pipe/kt',d['pipe_cyc_per_ktile'])"
  done
done
