T=tools/exp/conv_timeline.py
python $T --warm 200 --gemm --json gpurun_out/timeline_r02b_sustained.json > gpurun_out/timeline_r02b_sustained.txt 2>&1
for sh in 128,28,28,128,3,128,1,1 128,14,14,256,3,256,1,1 128,7,7,512,3,512,1,1; do
 EMBNET_CONV_TILE=0 python $T --warm 200 --shape $sh --only fwd >> gpurun_out/timeline_r02b_tile0.txt 2>&1
 EMBNET_CONV_TILE=1 python $T --warm 200 --shape $sh --only fwd >> gpurun_out/timeline_r02b_tile1.txt 2>&1
done
for sh in 128,56,56,64,3,64,1,1 128,28,28,128,3,128,1,1; do
 EMBNET_WGRAD_XCD=1 python $T --warm 200 --shape $sh --only wgrad >> gpurun_out/timeline_r02b_wxcd.txt 2>&1
done
grep -h kernel gpurun_out/timeline_r02b_*.txt | cut -c1-400
