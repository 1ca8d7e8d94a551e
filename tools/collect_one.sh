#!/bin/bash
# Refresh ONE config's committed measurements (gpurun -- 'bash tools/collect_one.sh rNN c1s'): its FETCH_SIZE / WRITE_SIZE
# passes merged into profiles/rNN_pmc_traffic.json, then its bench line + kernel table -> gpurun_out/<tag>/.
set -u
tag=${1:-r05}; c=${2:-c1s}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd "$root"
mkdir -p "$out"
cp "$root/profiles/${tag}_pmc_traffic.json" "$out/pmc_traffic.json"
PCMD="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-graph --sustain-seconds 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- $PCMD --config $c > /dev/null 2> "$out/pmc_fetch_$c.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- $PCMD --config $c > /dev/null 2> "$out/pmc_write_$c.err"
python3 tools/pmc_traffic.py "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_traffic.json" $c "$PCMD --config $c" > "$out/pmc_traffic_$c.txt"
rm -rf "$out/pmc_fetch" "$out/pmc_write"
cp "$out/pmc_traffic.json" "$root/profiles/${tag}_pmc_traffic.json"
python3 bench.py --config $c --steps 20 --warmup 5 --cpu-seconds 8 > "$out/bench_$c.json" 2> "$out/bench_$c.err"
head -c 600 "$out/bench_$c.json"; echo; tail -24 "$out/bench_$c.err"
