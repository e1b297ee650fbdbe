"""Which part of the data path costs the model step what?  Legs run as alternating windows inside ONE process
(process-to-process variation of the with-data leg is larger than the effects looked for):
  resident  : SAGE step on one resident batch, data path idle
  rotate    : on 8 resident batches in turn, data path idle
  data      : fed by the data path (what bench.py's model-step leg measures)
  decoupled : the data path runs at the same rate (one next() per step: chains + deliveries), but the model keeps
              training on the resident batch -- no dependency of the step on the delivered batch, no cold inputs
  gather    : resident batch + ONE lone row gather of 947 k rows per step on a side stream (no chains, no MFG export)
usage: overlap_ab.py [sage|gat] [rounds=4] [steps per window=64]"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import _native as nat  # noqa: E402
from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.shufflers import Shuffler  # noqa: E402
from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher  # noqa: E402
from salient_plusplus_amd.models import GAT, SAGE  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "sage"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 64
legs = os.environ.get("LEGS", "resident,rotate,data,decoupled,gather").split(",")
L = nat.load()
dev = torch.device("cuda", 0)
wl = make_workload(os.environ.get("WL", "S-papers"), seed=1234, device=dev)
slots = int(os.environ.get("SPP_MAX_SLOTS", "64"))
cfg = FastSamplerConfig(
    x_cpu=wl.x, x_gpu=torch.empty(0), y=wl.y.unsqueeze(-1), rowptr=wl.rowptr, col=wl.col, idx=wl.train_idx,
    batch_size=wl.batch_size, sizes=wl.fanouts, skip_nonfull_batch=False, pin_memory=False, distributed=False,
    partition_book=None, cache=fs.Cache(), force_exact_num_batches=True,
    exact_num_batches=max(1, wl.train_idx.numel() // wl.batch_size), count_remote_frequency=False, use_cache=False)
sampler = FastSampler(4, slots, cfg)
shuffler = Shuffler(wl.train_idx)


class Feeder:
    def __init__(self):
        self.epoch = 0
        self.it = None

    def next(self):
        while True:
            if self.it is None:
                shuffler.set_epoch(self.epoch)
                sampler.idx = shuffler.get_idx()
                self.it = DevicePrefetcher([dev], iter(sampler))
                self.epoch += 1
            b = next(self.it, None)
            if b is not None:
                return b[0]
            self.it = None


feeder = Feeder()
model = (GAT if arch == "gat" else SAGE)(wl.x.size(1), 256, 47, 3).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)


def step(b):
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.nll_loss(model(b.x, b.adjs), b.y.reshape(-1))
    loss.backward()
    opt.step()


for _ in range(3 * slots):
    fixed = feeder.next()
ring = [feeder.next() for _ in range(8)]
pos = [0]
N, F = wl.x.shape
U = 947_000
idx32 = torch.randint(0, N, (U,), device=dev, dtype=torch.int32)
gout = torch.empty((U, F), dtype=wl.x.dtype, device=dev)
side = torch.cuda.Stream(dev)


def leg_resident():
    step(fixed)


def leg_rotate():
    pos[0] = (pos[0] + 1) % len(ring)
    step(ring[pos[0]])


def leg_data():
    step(feeder.next())


def leg_decoupled():
    feeder.next()
    step(fixed)


def leg_gather():
    L.spp_gather_rows_strided(C.c_void_p(wl.x.data_ptr()), N, F * 2, wl.x.stride(0) * 2, C.c_void_p(idx32.data_ptr()), 4, U, U,
                              C.c_void_p(gout.data_ptr()), C.c_void_p(side.cuda_stream))
    step(fixed)


def capped(fn, wg):
    """the leg with the row gather / delivery grid capped at `wg` workgroups per compute unit"""
    def run():
        L.spp_tune(b"gather_wg_per_cu", wg)
        fn()
        L.spp_tune(b"gather_wg_per_cu", 16)
    return run


fns = {"resident": leg_resident, "rotate": leg_rotate, "data": leg_data, "decoupled": leg_decoupled, "gather": leg_gather}
for w in (1, 2, 4, 8):
    fns[f"gather_wg{w}"] = capped(leg_gather, w)
    fns[f"data_wg{w}"] = capped(leg_data, w)
res = {k: [] for k in legs}
host = {k: [] for k in legs}
insitu = {}
for r in range(rounds):
    for k in legs:
        fn = fns[k]
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
        L.spp_profile_enable(4)          # every 4th delivery launch and every chain timed with HIP events on their streams
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / steps * 1e3)
        host[k].append((t1 - t0) / steps * 1e3)
        for kind, name in ((0, "deliver/gather"), (2, "chain")):
            ms, n, u = C.c_double(0), C.c_int64(0), C.c_int64(0)
            L.spp_profile_read(kind, C.byref(ms), C.byref(n), C.byref(u))
            if n.value:
                insitu.setdefault((k, name), []).append(ms.value / n.value * 1e3)
        L.spp_profile_enable(0)
for k in legs:
    v = res[k]
    print(f"OVERLAP_AB {arch} {k:10s} mean {sum(v) / len(v):.4f} ms/step  windows " + " ".join(f"{x:.3f}" for x in v) +
          f"  | host enqueue ms/step {sum(host[k]) / len(host[k]):.3f}" +
          "".join(f"  | {name} in situ {sum(v2) / len(v2):.0f} us per launch" for (kk, name), v2 in insitu.items() if kk == k), flush=True)
