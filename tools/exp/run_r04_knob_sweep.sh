run() { echo -n "$* : "; env "$@" python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 --no-kernel-timer 2>/dev/null | sed "s/.*\"value\": \([0-9.]*\).*\"ms_per_step\": \([0-9.]*\).*/value \1 ms \2/"; }
run A=0
run EMBNET_WGRAD_NO192=1
run EMBNET_FAIR_SINGLE_ROUND=0
run EMBNET_TAIL_SLOTS=256
run EMBNET_TAIL_SLOTS=768
run A=0
run EMBNET_WGRAD_BLOCKS=768
run EMBNET_WGRAD_BLOCKS=1024
run EMBNET_WGRAD_BLOCKS=1536
run EMBNET_CONV_TAIL=0
run EMBNET_EPILOGUE_STATS=0
run A=0
run EMBNET_SLAB_DEFER=0
run EMBNET_WGRAD_TILE=0
run EMBNET_OVERLAP_WGRAD=1
run A=0
