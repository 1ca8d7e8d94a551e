# depthwise weight gradient, eight output columns per unit on stride-1 layers whose rows they tile (EMBNET_DW_WGRAD_TW8)
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 600 python -m pytest tests/test_mbconv_siamese_gpu.py tests/test_edge_cases_gpu.py -q -m gpu -x 2>&1 | tail -2
for v in 0 1; do echo "TW8=$v"; EMBNET_DW_WGRAD_TW8=$v python tools/exp/time_dw.py 2>/dev/null | sed "s/| fwd.*| wgrad dwconv_wgrad4_wave_kernel/wgrad/;s/: fwd.*| wgrad dwconv_wgrad4_wave_kernel/: wgrad/" | cut -c1-100; done
for i in 1 2 3; do
  one BCFG=c5 EMBNET_DW_WGRAD_TW8=0
  one BCFG=c5 EMBNET_DW_WGRAD_TW8=1
done
