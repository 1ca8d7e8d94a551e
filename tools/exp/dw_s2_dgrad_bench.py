import sys, os, json, torch
sys.path.insert(0, "/root/repo")
from embeddingnet_amd import _lib
from embeddingnet_amd._lib import check, stream
lib=_lib.lib(); dev=torch.device("cuda:0")
def timeit(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/iters*1e3
n=256
for (h,c,k) in [(14,672,5),(28,240,3),(56,144,5),(112,96,3)]:
    oh=h//2; pt=max((oh-1)*2+k-h,0)//2
    dys=[torch.randn(n,oh,oh,c,device=dev) for _ in range(4)]; es=[torch.randn(n,h,h,c,device=dev) for _ in range(4)]
    dxs=[torch.empty(n,h,h,c,device=dev) for _ in range(4)]
    w=torch.randn(k,k,c,1,device=dev); vec=[torch.rand(c,device=dev)+0.5 for _ in range(4)]
    rows=lib.embnet_dwconv2d_dgrad_bnsums_rows(n,h,h,c,k,k,2)
    st=torch.zeros(2,c,rows,device=dev); i=[0]
    def f():
        j=i[0]=(i[0]+1)%4
        check(lib.embnet_dwconv2d_dgrad_bnsums_f32(dys[j].data_ptr(),w.data_ptr(),dxs[j].data_ptr(),n,h,h,c,k,k,2,pt,pt,oh,oh,es[j].data_ptr(),vec[0].data_ptr(),vec[1].data_ptr(),vec[2].data_ptr(),vec[3].data_ptr(),2,st.data_ptr(),rows,stream()))
    t=min(timeit(f),timeit(f)); by=4.0*n*c*(2*h*h+oh*oh)
    print(json.dumps({"h":h,"c":c,"k":k,"us":round(t,1),"GBs":round(by/t/1e3)}))
