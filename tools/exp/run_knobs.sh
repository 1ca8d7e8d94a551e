for v in "" "EMBNET_CONV_TAIL=0" "EMBNET_FAIR_SINGLE_ROUND=0" "EMBNET_CONV_TAIL=0 EMBNET_FAIR_SINGLE_ROUND=0"; do
  for i in 1 2; do
    env $v python bench.py --no-cpu-baseline --steps 30 2>/dev/null | head -c 200 | tail -c 60 | tr -d '\n'; echo "   [$v]"
  done
done
