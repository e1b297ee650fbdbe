"""Registers / occupancy / LDS / spills of the kernels of one source file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py salient_plusplus_amd/csrc/sampler.hip [name filter ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
filters = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "salient_plusplus_amd", "csrc"), "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("SPP_EXTRA_FLAGS", "").split()
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
if not rows:
    sys.exit("no kernels found (compile error?):\n" + out[-2000:])
demangle = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'occ':>4s} {'LDS':>7s} {'scratch':>8s} {'vspill':>6s} {'sspill':>6s}")
for (mangled, r), name in zip(rows.items(), demangle):
    name = re.sub(r"\(.*", "", name).replace("spp::", "")
    if filters and not any(f in name for f in filters):
        continue
    print(f"{name[:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('SGPRs', '?'):>5s} {r.get('Occupancy [waves/SIMD]', '?'):>4s} "
          f"{r.get('LDS Size [bytes/block]', '?'):>7s} {r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('VGPRs Spill', '?'):>6s} {r.get('SGPRs Spill', '?'):>6s}")
