# thin 1x1 convs as an HBM stream (EMBNET_CONV_THIN): tests, then C5 A/B (the env is read once per process)
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 1200 python -m pytest tests/test_conv_thin_gpu.py tests/test_mbconv_siamese_gpu.py tests/test_reference_configs_gpu.py tests/test_fused_kernels_vs_oracle_gpu.py -q -x 2>&1 | tail -4
for i in 1 2 3; do
  one BCFG=c5 EMBNET_CONV_THIN=0
  one BCFG=c5 EMBNET_CONV_THIN=1
done
EMBNET_BENCH_ROWS=12 timeout 300 python bench.py --config c5 --steps 20 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep -E "thin|traced|conv_fwd|conv_dgrad"
