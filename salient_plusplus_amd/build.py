"""Builds libspp_hip.so (gfx950) in-tree with hipcc.  `python -m salient_plusplus_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libspp_hip.so")
SOURCES = ["api.hip", "mt19937.hip", "gather.hip", "sampler.hip", "partition.hip", "session.hip", "exchange.hip", "vip.hip", "aggregate.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
         "-I" + CSRC, "-Wall", "-Wno-unused-function"] + os.environ.get("SPP_EXTRA_FLAGS", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    # the flag set is part of what the objects were built from: an A/B script that compiled a variant with
    # SPP_EXTRA_FLAGS must not leave it behind for the next bench / test / profile on the same box
    stamp = os.path.join(OBJ, "flags.stamp")
    flags_now = " ".join([HIPCC] + FLAGS)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    headers += [os.path.join(ROOT, "include", "spp.h"), os.path.abspath(__file__)]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs + ["-ldl"])
    with open(stamp, "w") as f:
        f.write(flags_now)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
