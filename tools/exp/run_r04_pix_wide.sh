#!/bin/bash
# Squeeze-and-excite per-(image, channel) reductions (affine_act_gap4, se_bn_sums4): adaptive workgroup geometry
# (channel lanes = min(16, pow2 >= c4); 1024 threads where channel blocks x images < 2048) against the previous build.
# gpurun -- 'bash tools/exp/run_r04_pix_wide.sh'   (needs embeddingnet_amd/libembnet_hip_prev.so = the build before the change)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"
python -m pytest tests -x -q -m gpu -k "squeeze or se_gate or gate_multiply or pooled or mbconv or efficientnet" 2>&1 | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  EMBNET_LIB=$root/embeddingnet_amd/libembnet_hip_prev.so python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/pix_prev.err | line prev
  EMBNET_PIX_WIDE=0 python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/pix_0.err | line wide=0
  python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/pix_1.err | line wide=1
done
grep -h "affine_act_gap4\|se_bn_sums4" gpurun_out/pix_prev.err gpurun_out/pix_0.err gpurun_out/pix_1.err
