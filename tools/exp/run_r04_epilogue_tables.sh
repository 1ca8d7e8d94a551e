OLD=$PWD/build_variants/epi_old.so
for cfg in c5 c3; do
  for lib in $OLD ""; do
    tag=$( [ -z "$lib" ] && echo new || echo old )
    EMBNET_BENCH_ROWS=40 EMBNET_LIB=$lib BCFG=$cfg python bench.py --steps 12 --warmup 4 --no-cpu-baseline --sustain-seconds 0 > gpurun_out/abt_${cfg}_$tag.json 2> gpurun_out/abt_${cfg}_$tag.txt
  done
done
