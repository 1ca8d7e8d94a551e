for v in "EMBNET_TAIL_SLOTS=512" "EMBNET_TAIL_SLOTS=0" "EMBNET_TAIL_SLOTS=256" "EMBNET_TAIL_SLOTS=512"; do
  for i in 1 2; do
    env $v python bench.py --no-cpu-baseline --steps 30 2>/dev/null | head -c 200 | tail -c 60 | tr -d '\n'; echo "   [$v]"
  done
done
