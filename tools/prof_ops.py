"""Host (CPU) time per torch op of one SAGE training step fed by the data path (torch.profiler, CPU activity only):
where the ~0.8 ms of enqueue time per step goes (aten::mm: 6 calls of ~42 us each).  usage: prof_ops.py"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS","1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from salient_plusplus_amd import fast_sampler as fs
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher
from salient_plusplus_amd.models import SAGE
from salient_plusplus_amd.synthetic import make_workload
dev=torch.device("cuda",0)
wl=make_workload("S-papers",seed=1234,device=dev)
cfg=FastSamplerConfig(x_cpu=wl.x,x_gpu=torch.empty(0),y=wl.y.unsqueeze(-1),rowptr=wl.rowptr,col=wl.col,idx=wl.train_idx,batch_size=wl.batch_size,sizes=wl.fanouts,skip_nonfull_batch=False,pin_memory=False,distributed=False,partition_book=None,cache=fs.Cache(),force_exact_num_batches=True,exact_num_batches=wl.train_idx.numel()//wl.batch_size,count_remote_frequency=False,use_cache=False)
it=DevicePrefetcher([dev],iter(FastSampler(4,64,cfg)))
model=SAGE(wl.x.size(1),256,47,3).to(dev)
opt=torch.optim.Adam(model.parameters(),lr=1e-3,fused=True)
def step(b):
    opt.zero_grad(set_to_none=True)
    loss=torch.nn.functional.nll_loss(model(b.x,b.adjs),b.y.reshape(-1))
    loss.backward(); opt.step()
for _ in range(100): b=next(it)[0]; step(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(50):
        b=next(it)[0]; step(b)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
