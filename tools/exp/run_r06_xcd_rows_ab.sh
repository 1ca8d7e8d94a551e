#!/bin/bash
# A/B (same box, alternating): the column tiles of a patch-kernel tile row on one XCD (EMBNET_PATCH_XCD_ROWS) and the 256-row
# weight-gradient tile of the stem (EMBNET_WGRAD_NO256), C2; then the FETCH_SIZE / WRITE_SIZE passes of both placements.
out=gpurun_out/r06_exp_xcd_rows.txt
: > $out
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== C2 EMBNET_PATCH_XCD_ROWS=$v rep $rep" >> $out
    EMBNET_PATCH_XCD_ROWS=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batch-hard --sustain-seconds 0 2>&1 | grep -E "enqueue loop|conv_patch_kernel|traced kernels|\"metric\"" | cut -c1-230 >> $out
  done
done
for rep in 1 2; do
  for v in 1 0; do
    echo "== C2 EMBNET_WGRAD_256=$v rep $rep" >> $out
    EMBNET_WGRAD_256=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batch-hard --sustain-seconds 0 2>&1 | grep -E "enqueue loop|conv_wgrad_h_kernel|traced kernels|\"metric\"" | cut -c1-230 >> $out
  done
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PCMD="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-graph --no-batch-hard --sustain-seconds 0"
for v in 0 1; do
  o=gpurun_out/r06_xcd$v
  rm -rf $o; mkdir -p $o
  EMBNET_PATCH_XCD_ROWS=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/pmc_fetch -- $PCMD > /dev/null 2> $o/f.err
  EMBNET_PATCH_XCD_ROWS=$v rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/pmc_write -- $PCMD > /dev/null 2> $o/w.err
  python3 tools/pmc_traffic.py $o/pmc_fetch $o/pmc_write $o/pmc_traffic.json c2 "EMBNET_PATCH_XCD_ROWS=$v $PCMD" > gpurun_out/r06_pmc_traffic_c2_xcd$v.txt
  cp $o/pmc_traffic.json gpurun_out/r06_pmc_traffic_xcd$v.json
  rm -rf $o
done
