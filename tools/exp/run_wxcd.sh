for v in "0 0" "1 0" "1 1"; do
  set -- $v
  EMBNET_WGRAD_XCD=$1 EMBNET_WGRAD_STAGGER=$2 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/bench_wx_$1$2.json 2> gpurun_out/bench_wx_$1$2.err
  echo "xcd=$1 stagger=$2: $(head -c 230 gpurun_out/bench_wx_$1$2.json | tail -c 100)"
  grep "conv_wgrad" gpurun_out/bench_wx_$1$2.err | cut -c1-200
done
