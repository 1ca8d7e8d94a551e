set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r05_c3mfma
cd /tmp && export TMPDIR=/tmp && cd "$root"
rm -rf "$out" && mkdir -p "$out"
PCMD="python3 bench.py --config c3 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-graph --sustain-seconds 0"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$out/pmc_mfma" -- $PCMD > /dev/null 2> "$out/pmc_mfma.err"
python3 tools/pmc_mfma_clock.py "$out/pmc_mfma" "$out/pmc_mfma_clock_c3.md" "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES of \`$PCMD\`" > /dev/null
rm -rf "$out/pmc_mfma"
head -24 "$out/pmc_mfma_clock_c3.md"
