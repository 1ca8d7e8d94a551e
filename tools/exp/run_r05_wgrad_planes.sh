# weight gradient from the planes (EMBNET_WGRAD_PLANES): tests, then C2 and C3 with / without, alternating on one box
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 900 python -m pytest tests/test_wgrad_planes_gpu.py tests/test_conv_patch_gpu.py tests/test_round3_gpu.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2 3; do
  one BCFG=c2 EMBNET_WGRAD_PLANES=0
  one BCFG=c2 EMBNET_WGRAD_PLANES=1
done
for i in 1 2; do
  one BCFG=c3 EMBNET_WGRAD_PLANES=0
  one BCFG=c3 EMBNET_WGRAD_PLANES=1
done
