# depthwise weight gradient: channel-quad lane groups that fill the wave (EMBNET_DW_WGRAD_PACK) vs next power of two
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 600 python -m pytest tests/test_mbconv_siamese_gpu.py tests/test_edge_cases_gpu.py -q -m gpu -x 2>&1 | tail -2
for v in 0 1; do echo "EMBNET_DW_WGRAD_PACK=$v"; EMBNET_DW_WGRAD_PACK=$v python tools/exp/time_dw.py 2>/dev/null | sed "s/.*| wgrad/wgrad/" | cut -c1-120; done
for i in 1 2 3; do
  one BCFG=c5 EMBNET_DW_WGRAD_PACK=0
  one BCFG=c5 EMBNET_DW_WGRAD_PACK=1
done
