import torch, json
dev=torch.device("cuda:0")
def t(fn,it=20):
    for _ in range(3): fn()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e-3
for mb in (128, 512, 1200):
    n=mb*1024*1024//4
    a=[torch.empty(n,device=dev) for _ in range(3)]; b=[torch.empty(n,device=dev) for _ in range(3)]
    for x in a: x.normal_()
    i=[0]
    def fill():
        i[0]=(i[0]+1)%3; a[i[0]].fill_(1.0)
    def copy():
        i[0]=(i[0]+1)%3; b[i[0]].copy_(a[i[0]])
    def read():
        i[0]=(i[0]+1)%3; return a[i[0]].sum()
    def axpy():
        i[0]=(i[0]+1)%3; torch.add(a[i[0]], b[i[0]], out=b[(i[0]+1)%3])
    by=n*4
    print(json.dumps({"MB":mb,"fill_TBs":round(by/t(fill)/1e12,2),"copy_TBs":round(2*by/t(copy)/1e12,2),"read_TBs":round(by/t(read)/1e12,2),"add_TBs":round(3*by/t(axpy)/1e12,2)}))
