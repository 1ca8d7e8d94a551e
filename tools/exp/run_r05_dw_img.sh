# depthwise kernels: image-major thread order on small maps (EMBNET_DW_IMG_MAX), back to back per layer (tools/exp/time_dw.py)
for m in 0 256 1024 4096; do echo "== EMBNET_DW_IMG_MAX=$m"; EMBNET_DW_IMG_MAX=$m timeout 300 python tools/exp/time_dw.py 256 2>/dev/null | sed 's/| wgrad.*//'; done
