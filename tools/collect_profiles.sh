#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/collect_profiles.sh rNN'): bench lines of every BASELINE config, the rocprofv3
# kernel-trace summary of the headline command, the PMC passes for HBM traffic (separate passes, no tracing domains beside
# --kernel-trace) and for shader clock / MFMA utilisation -> gpurun_out/<tag>/.  Copy what should be judged into profiles/.
set -u
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd "$root"
rm -rf "$out" && mkdir -p "$out"
# HBM traffic counters first: the bench lines below look their dominant kernel up in profiles/rNN_pmc_traffic.json
PCMD="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-graph --no-batch-hard --sustain-seconds 0"
for c in c2 c1 c1s c3 c5; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- $PCMD --config $c > /dev/null 2> "$out/pmc_fetch_$c.err"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- $PCMD --config $c > /dev/null 2> "$out/pmc_write_$c.err"
  python3 tools/pmc_traffic.py "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_traffic.json" $c "$PCMD --config $c" > "$out/pmc_traffic_$c.txt"
  rm -rf "$out/pmc_fetch" "$out/pmc_write"
done
cp "$out/pmc_traffic.json" "$root/profiles/${tag}_pmc_traffic.json"
python3 bench.py > "$out/bench_c2.json" 2> "$out/bench_c2.err"
python3 bench.py --mining batch_hard --steps 30 --no-cpu-baseline > "$out/bench_c2_batch_hard.json" 2> "$out/bench_c2_batch_hard.err"
for c in c1 c1s c3 c5; do
  python3 bench.py --config $c --steps 20 --warmup 5 --cpu-seconds 8 > "$out/bench_$c.json" 2> "$out/bench_$c.err"
done
EMBNET_CONV_PATCH=0 python3 bench.py --no-cpu-baseline > "$out/bench_c2_gather_convs.json" 2> "$out/bench_c2_gather_convs.err"
CMD="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-batch-hard --sustain-seconds 0"
rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- $CMD > "$out/trace_bench.json" 2> "$out/trace.err"
python3 tools/kernel_stats.py "$out"/trace/*/*_kernel_trace.csv 13 "$out/kernel_stats" \
    "rocprofv3 --kernel-trace of \`$CMD\` (13 steps in the trace)" > /dev/null
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$out/pmc_mfma" -- $PCMD > /dev/null 2> "$out/pmc_mfma.err"
python3 tools/pmc_mfma_clock.py "$out/pmc_mfma" "$out/pmc_mfma_clock.md" "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES of \`$PCMD\`" > /dev/null
rm -rf "$out/pmc_mfma"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$out/pmc_mfma" -- $PCMD --config c3 > /dev/null 2> "$out/pmc_mfma_c3.err"
python3 tools/pmc_mfma_clock.py "$out/pmc_mfma" "$out/pmc_mfma_clock_c3.md" "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES of \`$PCMD --config c3\`" > /dev/null
rm -rf "$out/pmc_mfma" "$out"/trace/*/*agent_info.csv
head -c 900 "$out/bench_c2.json"; echo; tail -22 "$out/bench_c2.err"; head -14 "$out/kernel_stats.md"; head -8 "$out/pmc_traffic_c2.txt"; head -16 "$out/pmc_mfma_clock.md"
for c in c1 c1s c3 c5 c2_batch_hard c2_gather_convs; do head -c 400 "$out/bench_$c.json"; echo; tail -3 "$out/bench_$c.err"; done
