"""The model-step leg of bench.py, one leg per process, for a kernel trace:
LEG=resident : fwd + bwd + Adam of models.SAGE / GAT on ONE resident batch, the data path idle;
LEG=rotate   : the same with ROTATE (default 8) resident batches taken in turn, the data path idle: what the step costs
               when its inputs are not the ones it read a millisecond ago (x of a papers-scale batch is 242 MB, the
               Infinity Cache 256 MB);
LEG=data     : the same step fed by the data path (FastSampler -> DevicePrefetcher), i.e. the sampling
               chains and delivery launches of the next batches run beside the model's kernels.
Run each under `rocprofv3 --kernel-trace --output-format csv` and give the two CSVs to
tools/overlap_report.py.  usage: LEG=data python3 tools/overlap_trace.py [sage|gat] [steps=48] [workload=S-papers]"""
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.shufflers import Shuffler  # noqa: E402
from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher  # noqa: E402
from salient_plusplus_amd.models import GAT, SAGE  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "sage"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
leg = os.environ.get("LEG", "data")
dev = torch.device("cuda", 0)
wl = make_workload(sys.argv[3] if len(sys.argv) > 3 else "S-papers", seed=1234, device=dev)
slots = int(os.environ.get("SPP_MAX_SLOTS", "64"))
cfg = FastSamplerConfig(
    x_cpu=wl.x, x_gpu=torch.empty(0), y=wl.y.unsqueeze(-1), rowptr=wl.rowptr, col=wl.col, idx=wl.train_idx,
    batch_size=wl.batch_size, sizes=wl.fanouts, skip_nonfull_batch=False, pin_memory=False, distributed=False,
    partition_book=None, cache=fs.Cache(), force_exact_num_batches=True,
    exact_num_batches=max(1, wl.train_idx.numel() // wl.batch_size), count_remote_frequency=False, use_cache=False)
sampler = FastSampler(4, slots, cfg)
shuffler = Shuffler(wl.train_idx)
shuffler.set_epoch(0)
sampler.idx = shuffler.get_idx()
it = DevicePrefetcher([dev], iter(sampler))
model = (GAT if arch == "gat" else SAGE)(wl.x.size(1), 256, 47, 3).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)


def step(b):
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.nll_loss(model(b.x, b.adjs), b.y.reshape(-1))
    loss.backward()
    opt.step()


prime = 3 * slots if leg == "data" else 1
fixed = None
for _ in range(prime):            # allocator and workspace first-touch, as bench.py does
    fixed = next(it)[0]
if leg == "rotate":
    ring = [next(it)[0] for _ in range(int(os.environ.get("ROTATE", "8")))]
    pos = [0]

    def get():
        pos[0] = (pos[0] + 1) % len(ring)
        return ring[pos[0]]
else:
    get = (lambda: next(it)[0]) if leg == "data" else (lambda: fixed)
for _ in range(16):
    step(get())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(get())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"OVERLAP_TRACE leg={leg} {arch} {dt * 1e3:.4f} ms/step over {steps} steps ({wl.name if hasattr(wl, 'name') else ''})", flush=True)
