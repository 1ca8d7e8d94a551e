# depthwise weight gradient with the dy quads shared through LDS: A/B against the previous build (build_variants/dw_old.so)
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 600 python -m pytest tests/test_mbconv_siamese_gpu.py tests/test_edge_cases_gpu.py -q -m gpu -x 2>&1 | tail -2
OLD=$PWD/build_variants/dw_old.so
for lib in $OLD ""; do echo "lib=${lib:-new}"; EMBNET_LIB=$lib python tools/exp/time_dw.py 2>/dev/null | sed "s/| fwd.*| wgrad dwconv_wgrad4_wave_kernel/wgrad/;s/: fwd.*| wgrad dwconv_wgrad4_wave_kernel/: wgrad/" | cut -c1-100; done
for i in 1 2 3; do
  one BCFG=c5 EMBNET_LIB=$OLD
  one BCFG=c5
done
