import torch, time
dev=torch.device("cuda",0)
for n in (841_000*16, 841_000):
    t=torch.randint(0,1<<30,(n,),device=dev,dtype=torch.int32)
    idx=torch.randperm(n,device=dev)
    idx32=idx.to(torch.int32)
    out=torch.empty_like(t)
    for name,fn in (("t[idx] (int64 idx)", lambda: torch.index_select(t,0,idx,out=out)), ("take int32 idx", lambda: torch.index_select(t,0,idx32,out=out))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
        print(f"n={n}: {name}: {dt*1e6:.1f} us = {n/dt/1e9:.1f} G random 4-B reads/s")
