# full per-kernel tables of the C2 / C3 steps (library trace, 1 traced step) + every conv launch one by one
for c in c2 c3; do
  EMBNET_BENCH_ROWS=80 EMBNET_BENCH_DETAIL=conv timeout 900 python bench.py --config $c --steps 20 --warmup 6 --no-cpu-baseline --sustain-seconds 0 \
    > gpurun_out/r05_table_$c.json 2> gpurun_out/r05_table_$c.txt
  tail -3 gpurun_out/r05_table_$c.txt
done
