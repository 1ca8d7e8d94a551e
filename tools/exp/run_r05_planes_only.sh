# fp32 copies of planes tensors dropped where every reader takes the planes (EMBNET_PLANES_ONLY): tests, then C2 / C3 A/B
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 1200 python -m pytest tests/test_wgrad_planes_gpu.py tests/test_conv_patch_gpu.py tests/test_round3_gpu.py tests/test_backbone_gpu.py tests/test_step_parity_gpu.py -q -m gpu -x 2>&1 | tail -5
for i in 1 2 3; do
  one BCFG=c2 EMBNET_WGRAD_PLANES=0
  one BCFG=c2 EMBNET_PLANES_ONLY=0
  one BCFG=c2 EMBNET_PLANES_ONLY=1
done
