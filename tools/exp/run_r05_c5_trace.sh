#!/bin/bash
# rocprofv3 kernel-trace summary of the C5 bench command (profiles/r05_c5_kernel_stats.md).
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r05_c5_trace
cd /tmp && export TMPDIR=/tmp && cd "$root"
rm -rf "$out" && mkdir -p "$out"
CMD="python3 bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-graph --sustain-seconds 0"
rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- $CMD > "$out/trace_bench.json" 2> "$out/trace.err"
python3 tools/kernel_stats.py "$out"/trace/*/*_kernel_trace.csv 13 "$out/kernel_stats" "rocprofv3 --kernel-trace of \`$CMD\` (13 steps in the trace)" > /dev/null
rm -rf "$out/trace"
head -40 "$out/kernel_stats.md"
