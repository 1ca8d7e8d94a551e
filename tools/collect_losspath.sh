#!/bin/bash
# Loss-path evidence on the GPU box: `bash tools/collect_losspath.sh rNN` -> gpurun_out/<tag>/
#   gemm_sweep.json      distance GEMM, N x E grid, TFLOP/s against the fp32 MFMA peak (tools/kernel_bench.py gemm)
#   losspath_sweep.json  mining / hinge micro-benchmarks at N = 128..4096 (tools/kernel_bench.py losspath)
#   losspath.json        fused vs separate loss path at the BASELINE config sizes (tools/losspath_bench.py)
#   gemm_sweep_rocprofv3_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the sweep command
set -u
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd "$root"
mkdir -p "$out"
python3 tools/kernel_bench.py gemm --json "$out/gemm_sweep.json" > "$out/gemm_sweep.txt" 2>&1
python3 tools/kernel_bench.py losspath --json "$out/losspath_sweep.json" > "$out/losspath_sweep.txt" 2>&1
python3 tools/losspath_bench.py --json "$out/losspath.json" > "$out/losspath.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/gemm_trace" -- python3 tools/kernel_bench.py gemm > /dev/null 2> "$out/gemm_trace.err"
cp "$out"/gemm_trace/*/*kernel_stats.csv "$out/gemm_sweep_rocprofv3_kernel_stats.csv" 2>/dev/null
rm -rf "$out/gemm_trace"
cat "$out/gemm_sweep.txt"; tail -12 "$out/losspath.txt"
