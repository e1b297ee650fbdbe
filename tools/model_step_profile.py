"""One resident batch of the bench workload, N optimisation steps of models.SAGE / models.GAT on it
(fwd + bwd + Adam), timed; run under `rocprofv3 --kernel-trace --stats` for the per-kernel table.
usage: model_step_profile.py [sage|gat] [steps=30] [workload=S-papers]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher  # noqa: E402
from salient_plusplus_amd.models import GAT, SAGE  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "sage"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
wl = make_workload(sys.argv[3] if len(sys.argv) > 3 else "S-papers", seed=1234, device=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
cfg = FastSamplerConfig(
    x_cpu=wl.x, x_gpu=torch.empty(0), y=wl.y.unsqueeze(-1), rowptr=wl.rowptr, col=wl.col, idx=wl.train_idx[:8 * wl.batch_size],
    batch_size=wl.batch_size, sizes=wl.fanouts, skip_nonfull_batch=False, pin_memory=False, distributed=False,
    partition_book=None, cache=fs.Cache(), force_exact_num_batches=True, exact_num_batches=8,
    count_remote_frequency=False, use_cache=False)
it = DevicePrefetcher([dev], iter(FastSampler(2, 8, cfg)))
batch = next(it)[0]
torch.cuda.synchronize()
model = (GAT if arch == "gat" else SAGE)(wl.x.size(1), 256, 47, 3).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)    # one multi-tensor launch per step


def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.nll_loss(model(batch.x, batch.adjs), batch.y.reshape(-1))
    loss.backward()
    opt.step()
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"MODEL_STEP {arch} {dt * 1e3:.3f} ms/step on a resident batch: {batch.x.size(0)} nodes, "
      f"{[int(a.adj_t.nnz()) for a in batch.adjs]} edges, loss {float(loss):.4f}", flush=True)
