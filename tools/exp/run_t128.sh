for t in 384 3072; do
  EMBNET_CONV_T128_MIN=$t EMBNET_BENCH_DETAIL=conv_ python bench.py --no-cpu-baseline --steps 30 > gpurun_out/bench_t128_$t.json 2> gpurun_out/bench_t128_$t.err
  head -c 230 gpurun_out/bench_t128_$t.json | tail -c 100; echo
  grep "^    " gpurun_out/bench_t128_$t.err | grep -E "fwd|dgrad" | cut -c1-190
done
