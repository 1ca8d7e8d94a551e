# planes weight gradient: issue-priority knobs (EMBNET_WGP_KNOBS) and workgroup count, back to back on the ResNet layer sizes
for k in 0 1 2; do echo "== EMBNET_WGP_KNOBS=$k"; EMBNET_WGP_KNOBS=$k timeout 200 python tools/exp/wgrad_planes_bench.py 2>/dev/null | sed "s/'rel_diff.*//"; done
for b in 512 768; do echo "== EMBNET_WGRAD_PLANES_BLOCKS=$b"; EMBNET_WGRAD_PLANES_BLOCKS=$b timeout 200 python tools/exp/wgrad_planes_bench.py 2>/dev/null | sed "s/'rel_diff.*//"; done
