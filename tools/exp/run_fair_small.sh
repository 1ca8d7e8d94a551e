for sh in 40,56,56,64,3,64,1,1 80,28,28,128,3,128,1,1 41,56,56,64,3,64,1,1 128,56,56,64,3,64,1,1; do
python tools/exp/ab_conv.py build_variants/fair0.so build_variants/fair1.so --shapes $sh 2>&1 | grep -v amdgpu.ids | grep -v "^sum\|^shape"
done
