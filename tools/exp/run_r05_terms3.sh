#!/bin/bash
# What a THREE-term operand split would run at (fp16 x 2 pieces, three products): the product library against a variant
# built with -DEMBNET_EXP_TERMS=3 (tools/exp/_variants/terms3.so: the same kernels executing only the three largest of the six
# bf16 terms — NOT the product's arithmetic: 16-bit products; the pieces are still three planes, so operand traffic and LDS reads
# are unchanged).  C2 / C3 in the step, alternating.  Build first (CPU): bash tools/exp/run_r05_terms3.sh build
set -u
if [ "${1:-}" = "build" ]; then
  cd "$(dirname "$0")/../../embeddingnet_amd/csrc"
  out=../../tools/exp/_variants; mkdir -p $out/obj
  for f in *.hip; do /opt/rocm/bin/hipcc -DEMBNET_EXP_TERMS=3 -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -c $f -o $out/obj/${f%.hip}.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/terms3.so $out/obj/*.o && rm -rf $out/obj && echo built $out/terms3.so
  exit 0
fi
mkdir -p gpurun_out
O=gpurun_out/r05_exp_terms3.txt
: > $O
V=$(pwd)/tools/exp/_variants/terms3.so
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'loss', d['config'].get('loss_last_timed'), r['kernel'][:60], r['avg_us'])"; }
for r in 1 2 3; do
  echo "== c2 six terms (product) round=$r" >> $O
  BCFG=c2 timeout 300 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | line >> $O
  echo "== c2 three terms (variant) round=$r" >> $O
  BCFG=c2 EMBNET_LIB=$V EMBNET_LIB_LAX=1 timeout 300 python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | line >> $O
done
echo "== c3 six terms" >> $O
BCFG=c3 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | line >> $O
echo "== c3 three terms" >> $O
BCFG=c3 EMBNET_LIB=$V EMBNET_LIB_LAX=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | line >> $O
echo "== c2 three terms: kernel table" >> $O
BCFG=c2 EMBNET_LIB=$V EMBNET_LIB_LAX=1 timeout 300 python bench.py --steps 20 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep -E "x +[0-9]+/step|traced kernels" | head -8 >> $O
cat $O
