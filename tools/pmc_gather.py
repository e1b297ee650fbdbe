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
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
flush = torch.empty(1 << 28, dtype=torch.int32, device=dev)


def run(table_rows, idx, stride):
    x = torch.randn((table_rows, stride // 2), device=dev, dtype=torch.float16)
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
run(2_449_029, torch.randint(0, 2_449_029, (770_000,), device=dev, dtype=torch.int32, generator=g), STRIDE)
print("pmc_gather done", n_cal, 770_000, F, STRIDE)
