# depthwise weight gradient: units walked down a column block (L1 reuse of the image rows between the KS waves) vs row-major
one() { echo -n "$* : "; env "$@" timeout 900 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 600 python -m pytest tests/test_mbconv_siamese_gpu.py -q -m gpu -x 2>&1 | tail -2
for v in 0 1; do echo "EMBNET_DW_WGRAD_COLMAJOR=$v"; EMBNET_DW_WGRAD_COLMAJOR=$v python tools/exp/time_dw.py 2>/dev/null | cut -c1-200; done
for i in 1 2 3; do
  one BCFG=c5 EMBNET_DW_WGRAD_COLMAJOR=0
  one BCFG=c5 EMBNET_DW_WGRAD_COLMAJOR=1
done
