"""Workload for the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE) of the gather kernel:
  launch A: calibration -- sequential index over a table far larger than the Infinity Cache (every
            byte read and written exactly once: known traffic in THIS kernel's access width);
  launch B: the bench shape -- 770k random rows of 200 B out of the 2.45M-row S-products table,
            rows STRIDE bytes apart (256 = the resident layout padded to the fetch granule; 200 = dense).
Each launch is preceded by a 1 GiB fill so the caches are cold.  Development/measurement aid."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import _native as nat  # noqa: E402

L = nat.load()
nat.require_device()
dev = torch.device("cuda", 0)
F = int(os.environ.get("F", "100"))
STRIDE = int(os.environ.get("STRIDE", "256"))
TABLE_ROWS = int(os.environ.get("TABLE_ROWS", "2449029"))      # S-products; S-papers: 111059956
ROWS = int(os.environ.get("ROWS", "770000"))                   # rows per launch (S-papers: 947000)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
flush = torch.empty(1 << 28, dtype=torch.int32, device=dev)


def run(table_rows, idx, stride):
    x = torch.empty((table_rows, stride // 2), device=dev, dtype=torch.float16)
    x[:1 << 20].normal_()                                       # contents are irrelevant to the byte counters
    out = torch.empty((idx.numel(), F), dtype=torch.float16, device=dev)
    flush.fill_(1)
    torch.cuda.synchronize()
    nat.check(L.spp_gather_rows_strided(P(x), table_rows, F * 2, stride, P(idx), 4, idx.numel(), idx.numel(), P(out),
                                        st))
    torch.cuda.synchronize()


n_cal = 4_000_000
run(n_cal, torch.arange(n_cal, device=dev, dtype=torch.int32), F * 2)      # calibration: dense, every byte once
g = torch.Generator(device=dev)
g.manual_seed(1)
run(TABLE_ROWS, torch.randint(0, TABLE_ROWS, (ROWS,), device=dev, dtype=torch.int32, generator=g), STRIDE)
print("pmc_gather done", n_cal, ROWS, F, STRIDE, TABLE_ROWS)
