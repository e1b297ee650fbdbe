"""The row gather ALONE at papers shape (947 k random rows of 256 B out of a 28 GB table) under a workgroup cap.
usage: python3 tools/gather_occupancy.py     (GPU box; SPP_GATHER_WG_PER_CU is read once per process)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from salient_plusplus_amd import _native as nat  # noqa: E402

L = nat.load()
dev = torch.device("cuda", 0)
N, F, U = 111_059_956, 128, 947_000
x = torch.empty((N, F), dtype=torch.float16, device=dev)
idx = torch.randint(0, N, (U,), device=dev, dtype=torch.int32)
out = torch.empty((U, F), dtype=torch.float16, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for _ in range(50):
    L.spp_gather_rows(P(x), N, F * 2, P(idx), 4, U, U, P(out), st)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 300
for _ in range(n):
    L.spp_gather_rows(P(x), N, F * 2, P(idx), 4, U, U, P(out), st)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / n * 1e6
print(f"wg_per_cu={os.environ.get('SPP_GATHER_WG_PER_CU', '16')}: {us:.1f} us = {U * 520 / us / 1e6:.2f} TB/s algorithmic", flush=True)
