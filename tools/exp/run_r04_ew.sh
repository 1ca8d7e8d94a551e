one() { echo -n "$* : "; env "$@" python bench.py --steps 30 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
for i in 1 2; do for v in 4096 1024 2048 8192 16384; do one EMBNET_EW_BLOCKS=$v BCFG=c2; done; done
for v in 4096 2048 8192; do one EMBNET_EW_BLOCKS=$v BCFG=c5; done
