for i in 1 2; do
  echo shipped; python tools/exp/time_patch.py 128 2>/dev/null | cut -c1-80
  echo plain-epilogue; EMBNET_LIB=$PWD/build_variants/patch_plain.so python tools/exp/time_patch.py 128 2>/dev/null | cut -c1-80
done
