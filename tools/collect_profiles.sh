#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/collect_profiles.sh rNN'): the headline bench line, the rocprofv3
# kernel-trace summary of the same command, and the two PMC passes for HBM traffic -> gpurun_out/<tag>/.
# Copy the results you want judged into profiles/ afterwards (tools/collect_profiles.sh only writes scratch).
set -u
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd "$root"
rm -rf "$out" && mkdir -p "$out"
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline \
    > "$out/trace_bench.json" 2> "$out/trace.err"
python3 tools/kernel_stats.py "$out"/trace/*/*_kernel_trace.csv 13 "$out/kernel_stats" \
    "rocprofv3 --kernel-trace of \`python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline\` (13 steps in the trace)" > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 4 --warmup 2 \
    --no-cpu-baseline --no-kernel-timer > /dev/null 2> "$out/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 4 --warmup 2 \
    --no-cpu-baseline --no-kernel-timer > /dev/null 2> "$out/pmc_write.err"
python3 tools/pmc_traffic.py "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_traffic.json" > "$out/pmc_traffic.txt"
rm -rf "$out/pmc_fetch" "$out/pmc_write" "$out"/trace/*/*agent_info.csv
head -c 600 "$out/bench.json"; echo; head -12 "$out/kernel_stats.md"; head -5 "$out/pmc_traffic.txt"
