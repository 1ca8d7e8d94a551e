one() { echo -n "$* : "; env "$@" BCFG=c5 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
one X=0
one EMBNET_PAD_INPUT_CONV=0
one EMBNET_CONV_TILE=1
one EMBNET_CONV_TILE=3
one X=0
one EMBNET_CONV_T128_MIN=100000
one EMBNET_SMALL_SPLIT=0
