"""How well does the delivery's row gather co-run with the model step's kernels?  
For each model-side candidate A (the layer-1 forward GEMM, its weight-gradient GEMM, the fp16 mean
aggregation, a whole resident SAGE step) and the data-side kernel B (the row gather of a batch's 947 k
rows out of the 28 GB table): time of nA launches of A alone, nB launches of B alone, and of both
enqueued together on two streams.  hidden = (tA + tB - tAB) / min(tA, tB): 1 = the shorter side
disappears under the longer one, 0 = they add up.
usage: corun.py [workload=S-papers]      (env SPP_GATHER_WG_PER_CU caps the gather's occupancy)"""
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
from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher  # noqa: E402
from salient_plusplus_amd.models import SAGE  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

L = nat.load()
dev = torch.device("cuda", 0)
wl = make_workload(sys.argv[1] if len(sys.argv) > 1 else "S-papers", seed=1234, device=dev)
N, F = wl.x.shape
U = 947_000
idx32 = torch.randint(0, N, (U,), device=dev, dtype=torch.int32)
out = torch.empty((U, F), dtype=wl.x.dtype, device=dev)
side = torch.cuda.Stream(dev)


def P(t):
    return C.c_void_p(t.data_ptr())


def gather():
    L.spp_gather_rows_strided(P(wl.x), N, F * 2, wl.x.stride(0) * 2, P(idx32), 4, U, U, P(out), C.c_void_p(side.cuda_stream))


# model-side candidates on the current stream
T1 = 180_224
a1 = torch.randn(T1, 256, device=dev)
w1 = torch.randn(256, 256, device=dev)
g1 = torch.randn(T1, 256, device=dev)
cfg = FastSamplerConfig(
    x_cpu=wl.x, x_gpu=torch.empty(0), y=wl.y.unsqueeze(-1), rowptr=wl.rowptr, col=wl.col, idx=wl.train_idx[:8 * wl.batch_size],
    batch_size=wl.batch_size, sizes=wl.fanouts, skip_nonfull_batch=False, pin_memory=False, distributed=False,
    partition_book=None, cache=fs.Cache(), force_exact_num_batches=True, exact_num_batches=8,
    count_remote_frequency=False, use_cache=False)
it = DevicePrefetcher([dev], iter(FastSampler(2, 8, cfg)))
batch = next(it)[0]
torch.cuda.synchronize()
model = SAGE(F, 256, 47, 3).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)


def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.nll_loss(model(batch.x, batch.adjs), batch.y.reshape(-1))
    loss.backward()
    opt.step()


def fwd_only():
    with torch.no_grad():
        model(batch.x, batch.adjs)


cands = [
    ("GEMM fwd layer 1 (180k x 256 @ 256 x 256, fp32)", lambda: torch.mm(a1, w1.t()), 40),
    ("GEMM weight grad (256 x 180k @ 180k x 256, fp32)", lambda: torch.mm(g1.t(), a1), 40),
    ("elementwise add (HBM bound, 2 x 184 MB read + 184 MB write)", lambda: torch.add(a1, g1), 40),
    ("SAGE forward only (resident batch)", fwd_only, 12),
    ("SAGE step fwd+bwd+Adam (resident batch)", step, 12),
]


def timed(fa, na, fb, nb):
    """wall time (ms) of na launches of fa on the current stream and nb of fb on the side stream, interleaved"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ia = ib = 0
    while ia < na or ib < nb:
        if ia < na and (nb == 0 or ib >= nb or ia * nb <= ib * na):
            fa()
            ia += 1
        else:
            fb()
            ib += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(5):
    gather()
    step()
torch.cuda.synchronize()
tg1 = timed(None, 0, gather, 50) / 50
print(f"gather alone: {tg1 * 1e3:.1f} us per launch ({U * (4 * F + 4) / tg1 / 1e6:.0f} GB/s algorithmic), "
      f"SPP_GATHER_WG_PER_CU={os.environ.get('SPP_GATHER_WG_PER_CU', '16 (default)')}", flush=True)
for name, fa, na in cands:
    for _ in range(3):
        fa()
    ta1 = timed(fa, na, None, 0) / na
    # as many gathers as fill the same time, and the fixed 1-per-step ratio of the pipeline
    for nb in (max(1, round(na * ta1 / tg1)), max(1, round(na * ta1 / 1.0)) if "step" in name else 0):
        if nb == 0:
            continue
        ta = timed(fa, na, None, 0)
        tb = timed(None, 0, gather, nb)
        tab = timed(fa, na, gather, nb)
        hidden = (ta + tb - tab) / min(ta, tb)
        print(f"{name}: A x{na} {ta:.3f} ms, gather x{nb} {tb:.3f} ms, together {tab:.3f} ms -> hidden {hidden:+.2f} "
              f"(cost of a gather beside A: {(tab - ta) / nb * 1e3:.1f} us, alone {tb / nb * 1e3:.1f})", flush=True)
