# one-launch BatchNorm backward (bn_bwd_fused4_kernel) inside the step: `bash tools/exp/run_r04_bn_fused.sh`
one() { echo -n "$* : "; env "$@" timeout 600 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_round4_gpu.py -q -m gpu -k "batchnorm or relu_backward" 2>&1 | tail -5
for i in 1 2; do
  for cfg in c1 c1s c2 c5; do
    one EMBNET_BN_FUSED_MAX=0 BCFG=$cfg
    one EMBNET_BN_FUSED_MAX=2097152 BCFG=$cfg
    one EMBNET_BN_FUSED_MAX=8388608 BCFG=$cfg
    one EMBNET_BN_FUSED_MAX=33554432 BCFG=$cfg
  done
done
