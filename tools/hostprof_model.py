"""Host time of one optimisation step of models.SAGE, resident batch vs fed by the data path, with the GPU queue empty
before every step (no back-pressure in the figures): whole step, then piece by piece (next / zero_grad / forward / loss /
backward / optimizer); PROFILE=1 adds a cProfile of each leg.  usage: hostprof_model.py [steps=200]"""
import cProfile
import os
import pstats
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher  # noqa: E402
from salient_plusplus_amd.models import SAGE  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
wl = make_workload(os.environ.get("WL", "S-papers"), seed=1234, device=dev)
cfg = FastSamplerConfig(
    x_cpu=wl.x, x_gpu=torch.empty(0), y=wl.y.unsqueeze(-1), rowptr=wl.rowptr, col=wl.col, idx=wl.train_idx,
    batch_size=wl.batch_size, sizes=wl.fanouts, skip_nonfull_batch=False, pin_memory=False, distributed=False,
    partition_book=None, cache=fs.Cache(), force_exact_num_batches=True,
    exact_num_batches=max(1, wl.train_idx.numel() // wl.batch_size), count_remote_frequency=False, use_cache=False)
it = DevicePrefetcher([dev], iter(FastSampler(4, 64, cfg)))
model = SAGE(wl.x.size(1), 256, 47, 3).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)


def step(b):
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.nll_loss(model(b.x, b.adjs), b.y.reshape(-1))
    loss.backward()
    opt.step()


for _ in range(200):
    fixed = next(it)[0]
    step(fixed)


def leg(get, n, prof=None):
    tot_get = tot_step = 0.0
    for _ in range(n):
        torch.cuda.synchronize()
        if prof:
            prof.enable()
        t0 = time.perf_counter()
        b = get()
        t1 = time.perf_counter()
        step(b)
        t2 = time.perf_counter()
        if prof:
            prof.disable()
        tot_get += t1 - t0
        tot_step += t2 - t1
    return tot_get / n * 1e6, tot_step / n * 1e6


for name, get in (("resident", lambda: fixed), ("data", lambda: next(it)[0])):
    leg(get, 20)
    g, s = leg(get, steps)
    print(f"HOSTPROF {name}: next() {g:.1f} us, step() {s:.1f} us of host time per step (queue empty before each step)", flush=True)
def pieces(get, n):
    acc = [0.0] * 6
    for _ in range(n):
        torch.cuda.synchronize()
        t = [time.perf_counter()]
        b = get()
        t.append(time.perf_counter())
        opt.zero_grad(set_to_none=True)
        t.append(time.perf_counter())
        out = model(b.x, b.adjs)
        t.append(time.perf_counter())
        loss = torch.nn.functional.nll_loss(out, b.y.reshape(-1))
        t.append(time.perf_counter())
        loss.backward()
        t.append(time.perf_counter())
        opt.step()
        t.append(time.perf_counter())
        for k in range(6):
            acc[k] += t[k + 1] - t[k]
    return [v / n * 1e6 for v in acc]


for name, get in (("resident", lambda: fixed), ("data", lambda: next(it)[0])):
    pieces(get, 20)
    v = pieces(get, steps)
    print(f"HOSTPIECES {name}: next {v[0]:.1f}  zero_grad {v[1]:.1f}  forward {v[2]:.1f}  loss {v[3]:.1f}  backward {v[4]:.1f}  "
          f"optimizer {v[5]:.1f} us", flush=True)
del fixed
import gc
gc.collect()
for name, get in (("data, previous batch dropped before the step", lambda: next(it)[0]),):
    v = pieces(get, steps)
    print(f"HOSTPIECES {name}: next {v[0]:.1f}  zero_grad {v[1]:.1f}  forward {v[2]:.1f}  loss {v[3]:.1f}  backward {v[4]:.1f}  "
          f"optimizer {v[5]:.1f} us", flush=True)
if os.environ.get("PROFILE") != "1":
    sys.exit(0)
fixed = next(it)[0]
for name, get in (("resident", lambda: fixed), ("data", lambda: next(it)[0])):
    pr = cProfile.Profile()
    leg(get, steps, pr)
    print(f"---- cProfile {name} ({steps} steps) ----")
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
