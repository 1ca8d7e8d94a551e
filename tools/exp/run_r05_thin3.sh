# thin 1x1 kernels with branch-free prefetch: tests, then C5: everything off / forward+dgrad only / + weight gradient
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 600 python -m pytest tests/test_conv_thin_gpu.py -q -x 2>&1 | grep -E "Error|assert|passed|failed" | head -8
for i in 1 2 3; do
  one BCFG=c5 EMBNET_CONV_THIN=0
  one BCFG=c5 EMBNET_CONV_THIN_WGRAD=0
  one BCFG=c5 EMBNET_CONV_THIN_WGRAD=1
done
EMBNET_BENCH_ROWS=8 EMBNET_BENCH_DETAIL=thin timeout 300 python bench.py --config c5 --steps 20 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep -E "thin|traced"
