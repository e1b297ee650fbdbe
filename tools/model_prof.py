"""Kernel profile of the SAGE fallback step on one resident batch (development aid).
usage: rocprofv3 --kernel-trace --stats --output-format csv -d out -o m -- python3 tools/model_prof.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher  # noqa: E402
from salient_plusplus_amd.models import GAT, SAGE  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

dev = torch.device("cuda", 0)
wl = make_workload(os.environ.get("WL", "S-products"), device=dev)
cfg = FastSamplerConfig(
    x_cpu=wl.x, x_gpu=torch.empty(0), y=wl.y.unsqueeze(-1), rowptr=wl.rowptr, col=wl.col, idx=wl.train_idx[:8192],
    batch_size=wl.batch_size, sizes=wl.fanouts, skip_nonfull_batch=False, pin_memory=False, distributed=False,
    partition_book=None, cache=fs.Cache(), force_exact_num_batches=True, exact_num_batches=8,
    count_remote_frequency=False, use_cache=False)
(b,) = next(iter(DevicePrefetcher([dev], iter(FastSampler(2, 8, cfg)))))
torch.cuda.synchronize()
model = (GAT if os.environ.get('MODEL') == 'gat' else SAGE)(wl.x.size(1), 256, 47, 3).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
print("batch:", b.x.shape, [tuple(a.size) for a in b.adjs], flush=True)
for _ in range(int(os.environ.get("STEPS", "20"))):
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.nll_loss(model(b.x, b.adjs), b.y.reshape(-1))
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("done", float(loss.detach()))
