"""Which formulation of the SAGE layer GEMMs does the library run fastest? (fp32; default M=164k, K=256, N=256 = layer 1;
MKN=15333,512,256 = layer 2)"""
import os

import torch

dev = torch.device("cuda", 0)
M, K, N = (int(v) for v in os.environ.get("MKN", "164000,256,256").split(","))
a = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev)
g = torch.randn(M, N, device=dev)
wt = w.t().contiguous()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print("forward  a @ w.t():            %.1f us" % timeit(lambda: a @ w.t()))
print("forward  a @ wt_contig:        %.1f us" % timeit(lambda: a @ wt))
print("forward  F.linear(a, w):       %.1f us" % timeit(lambda: torch.nn.functional.linear(a, w)))
print("forward  (w @ a.t()).t():      %.1f us" % timeit(lambda: (w @ a.t()).t()))
for parts in (2, 4, 8):
    c = M // parts
    out = torch.empty(M, N, device=dev)

    def f():
        for i in range(parts):
            torch.mm(a[i * c:(i + 1) * c], wt, out=out[i * c:(i + 1) * c])
    print(f"forward  {parts} row chunks (mm out=): %.1f us" % timeit(f))
print("wgrad    g.t() @ a:            %.1f us" % timeit(lambda: g.t() @ a))
for slabs in (2, 4, 8, 16, 32, 64, 128, 256):
    c = M // slabs
    if c < 256:
        continue

    def f():
        return torch.bmm(g[:slabs * c].view(slabs, c, N).transpose(1, 2), a[:slabs * c].view(slabs, c, K)).sum(0)
    print(f"wgrad    bmm {slabs:3d} slabs + sum:   %.1f us" % timeit(f))

    def f2():
        return torch.bmm(a[:slabs * c].view(slabs, c, K).transpose(1, 2), g[:slabs * c].view(slabs, c, N)).sum(0).t()
    print(f"wgrad    bmm {slabs:3d} slabs (a^T g) + sum: %.1f us" % timeit(f2))
# reduced precision references (NOT used: the reference trains in fp32)
a16, w16 = a.half(), w.half()
print("(fp16)   a16 @ w16.t():        %.1f us" % timeit(lambda: a16 @ w16.t()))
