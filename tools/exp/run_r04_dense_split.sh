one() { echo -n "$* : "; env "$@" python bench.py --steps 15 --warmup 4 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
for i in 1 2; do for v in 64 32 16 8; do one EMBNET_DENSE_SPLIT_MIN_KT=$v BCFG=c5; done; done
for v in 64 16; do one EMBNET_DENSE_SPLIT_MIN_KT=$v BCFG=c1; one EMBNET_DENSE_SPLIT_MIN_KT=$v BCFG=c2; done
