"""Which formulation of the layer-1 forward GEMM (180 k x 256 @ 256 x 256, fp32) tolerates a row gather beside it best?
Per formulation: time alone; time of N launches with back-to-back gathers of 947 k rows on a side stream; the cost of one
gather beside it (alone: ~78 us).  usage: corun_gemm.py"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import _native as nat  # noqa: E402

L = nat.load()
dev = torch.device("cuda", 0)
N_ROWS, F = 60_000_000, 128          # a 15 GB fp16 table: rows far apart, as in the 28 GB papers table
x = torch.empty((N_ROWS, F), dtype=torch.float16, device=dev).normal_()
U = 947_000
idx32 = torch.randint(0, N_ROWS, (U,), device=dev, dtype=torch.int32)
out_g = torch.empty((U, F), dtype=torch.float16, device=dev)
side = torch.cuda.Stream(dev)


def gather():
    L.spp_gather_rows_strided(C.c_void_p(x.data_ptr()), N_ROWS, F * 2, F * 2, C.c_void_p(idx32.data_ptr()), 4, U, U,
                              C.c_void_p(out_g.data_ptr()), C.c_void_p(side.cuda_stream))


M, K, N = 180_224, 256, 256
a = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev)
wt = w.t().contiguous()
out = torch.empty(M, N, device=dev)


def chunks(parts):
    c = M // parts

    def f():
        for i in range(parts):
            torch.mm(a[i * c:(i + 1) * c], wt, out=out[i * c:(i + 1) * c])
    return f


def col_split(parts):
    c = N // parts

    def f():
        for i in range(parts):
            torch.mm(a, wt[:, i * c:(i + 1) * c], out=None)
    return f


cands = [("a @ w.t()", lambda: torch.mm(a, w.t())), ("a @ wt (contiguous)", lambda: torch.mm(a, wt)),
         ("F.linear", lambda: torch.nn.functional.linear(a, w)), ("(w @ a.t()).t()", lambda: torch.mm(w, a.t()))]
for p in (2, 4, 8, 16, 32, 64):
    cands.append((f"{p} row chunks", chunks(p)))
for p in (2, 4):
    cands.append((f"{p} column slices", col_split(p)))


def timed(fa, na, nb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ia = ib = 0
    while ia < na or ib < nb:
        if ia < na and (nb == 0 or ib >= nb or ia * nb <= ib * na):
            fa()
            ia += 1
        else:
            gather()
            ib += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(5):
    gather()
torch.cuda.synchronize()
tg = timed(None, 0, 60) / 60
print(f"gather alone {tg * 1e3:.1f} us")
for name, fa in cands:
    for _ in range(3):
        fa()
    na = 40
    ta = timed(fa, na, 0)
    nb = max(1, round(ta / tg))
    tb = timed(None, 0, nb)
    tab = timed(fa, na, nb)
    print(f"{name:22s} alone {ta / na * 1e3:7.1f} us | x{na} + {nb} gathers: {tab:.3f} ms vs {ta:.3f} + {tb:.3f} -> hidden "
          f"{(ta + tb - tab) / min(ta, tb):+.2f}, a gather costs {(tab - ta) / nb * 1e3:.1f} us beside it", flush=True)
