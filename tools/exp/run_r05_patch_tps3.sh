# patch kernel: three taps per barrier at BN = 128 too (EMBNET_PATCH_TPS3=1: <128,3,3,3,2>, two 36 KB weight slots) — tests, then C2 A/B
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
EMBNET_PATCH_TPS3=1 timeout 900 python -m pytest tests/test_conv_patch_gpu.py -q -x 2>&1 | tail -2
for i in 1 2 3; do
  one BCFG=c2 EMBNET_PATCH_TPS3=0
  one BCFG=c2 EMBNET_PATCH_TPS3=1
done
EMBNET_PATCH_TPS3=1 EMBNET_BENCH_ROWS=6 timeout 300 python bench.py --steps 20 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep "conv_patch"
