#!/usr/bin/env python3
"""Hot-path benchmark: sampled-edges/s of the mini-batch data pipeline (neighbour sampling ->
MFG -> feature/label slice -> PreparedBatch in HBM) on the dataset BASELINE.json's metric is quoted
on: ogbn-papers100M scale (synthetic, seeded: 111 M nodes, 3.2 G symmetric nnz, F=128 fp16 -- 56 GB
of topology + features, which fit one MI355X), GraphSAGE fanout [15,10,5], batch 1024.  The graph
carries a planted 8-block locality (80 % intra-block edges) as the stand-in for the METIS partitions
BASELINE.json's multi-GPU configurations name (synthetic.py: LOCALITY; `--workload S-papers-uniform`
is the same graph without it -- identical single-GPU numbers, 7/8 of all neighbours remote on 8 GPUs).
`--workload S-products` runs BASELINE.json's configs[1] instead.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: this script starts
                                                           its N rank processes itself, as children)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one batch through the hot path.  N == 1: single-GPU iterator (FastSampler +
DevicePrefetcher), all features in HBM.  N > 1: one process per GPU, replicated topology, features
range-partitioned N ways with a VIP cache of remote rows (analytic model, 10 % of N/P), cache-miss
rows fetched over RCCL/xGMI by the native exchange (DeviceDistributedPrefetcher); every rank samples
from the training vertices of its own partition (the reference launcher's "federated" default) and
runs K batches (weak scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import ctypes as C
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# As the reference's launcher does (utils/exp_driver.py:154 exports OMP_NUM_THREADS=1).  On the GPU boxes
# nproc is 256 while the cgroup grants 16 cores: torch's intra-op OpenMP pool then spins 256 threads after
# every small CPU op of an epoch boundary (the seeded CPU randperm of the shuffler), exhausts the CPU quota
# and the whole process is throttled for the rest of the 100 ms period -- a 75 ms stall with an idle GPU
# every few epochs (tools/epoch_boundary.py: 9 of 24 S-arxiv epochs took 80 ms instead of 7).
os.environ.setdefault("OMP_NUM_THREADS", "1")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=192)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default=os.environ.get("SPP_BENCH_WORKLOAD", "S-papers"),
                    help="S-papers (default: the dataset BASELINE.json's metric is quoted on; 56 GB of graph + "
                         "features fit one MI355X), S-products (configs[1]), S-arxiv, S-tiny")
    ap.add_argument("--slots", type=int, default=0,
                    help="batch slots in flight (0 = 64: 4 slot-sets of 16)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target length of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-model-step", action="store_true",
                    help="skip the (untimed-in-value) SAGE step measurement that gives the epoch-time figure")
    ap.add_argument("--cache-frac", type=float, default=0.10)
    ap.add_argument("--model", default="sage", choices=["sage", "gat"],
                    help="consumer model of the epoch-time figure (models.py); the plain-torch comparison is SAGE")
    ap.add_argument("--no-verify", action="store_true",
                    help="N>1: skip the bit-exact check of one group of batches through the native exchange")
    ap.add_argument("--prime", type=int, default=-1,
                    help="steps run as part of set-up, before the W warm-up steps: first-touch costs of the "
                         "allocators, the pooled sampler workspace and the exchange buffers (reported as priming_steps; "
                         "-1 = 3 x the slots in flight: the consumer loop of this benchmark never waits for the GPU, so it "
                         "runs up to `slots` deliveries ahead of it and torch's caching allocator keeps growing by one "
                         "output batch per step until that depth is reached -- at MAG240 scale one of those growths took "
                         "0.5 s, profiles/r03_bench_mag_stall.txt)")
    ap.add_argument("--cache-strategy", default="vip", choices=["vip", "degree", "degree-desc"],
                    help="N>1: ranking of the remote vertices for the feature cache (ddp.py:425-492)")
    ap.add_argument("--seed-scheme", default="federated", choices=["federated", "global"],
                    help="N>1: each rank trains on the training vertices of its own partition (the reference "
                         "launcher's default) or on a 1/N slice of one global permutation")
    ap.add_argument("--windows", type=int, default=0,
                    help="timed windows of --steps steps run back to back, pipeline kept full in between; the "
                         "reported ms_per_step / value are the plain mean over the windows "
                         "(0 = max(6, 2 * ceil(128 / steps)))")
    ap.add_argument("--fanouts", default="",
                    help="comma-separated fan-outs instead of the workload's own (e.g. 20,20,20: the reference's batchwise "
                         "inference; a hop < 0 or > 32 takes the generic sampling path: one batch per launch, a host sync per hop)")
    ap.add_argument("--no-fused-leg", action="store_true",
                    help="N=1, SAGE: skip the model-step leg of the opt-in fused consumer (Session(table_features) + "
                         "models.SAGE reading its first layer straight from the resident feature table)")
    ap.add_argument("--cpu-cores", type=int, default=0,
                    help="confine the process to this many cores before torch is imported (0 = no confinement): the host "
                         "budget of one rank of an 8-rank node (16-core quota / 8 = 2) rehearsed on one GPU; not under rocprofv3")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="launcher only: print the GPU count found without the HIP runtime and whether libamdhip64 is mapped "
                         "into the launcher process, start nothing")
    ap.add_argument("--hidden", type=int, default=256, help="hidden width of the consumer model (BASELINE configs[4]: 1024)")
    ap.add_argument("--layers", type=int, default=0,
                    help="layers of the consumer model (0 = one per hop of the fan-out list, what the MFG can feed: "
                         "driver/models.py:41-50 consumes one adj per layer)")
    ap.add_argument("--epochs", type=int, default=4,
                    help="WHOLE epochs timed wall-clock per leg (iterator creation -> last batch -> synchronize), the first "
                         "reported separately as the reference does (fast_trainer/train.py:223-316 drops it); 0 = skip")
    ap.add_argument("--p2p-leg", action="store_true",
                    help="partitioned path: after the default (RCCL) windows, a second short leg over the opt-in P2P transport "
                         "(SPP_DIST_TRANSPORT=p2p: remote rows read in their owners' partitions, mapped through HIP IPC); its "
                         "figures are appended under `exchange_p2p`, the line's value stays the RCCL one")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the partitioned / RCCL exchange path even with one rank (rehearsal of the N>1 code)")
    return ap.parse_args()


def host_cpu_share() -> int:
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def visible_gpu_count() -> int:
    """GPUs this process could open, counted WITHOUT the HIP runtime (the launcher process below must never
    initialise it: it forks the rank processes): the KFD topology nodes that have SIMDs (CPU nodes have none) and
    whose DRM render node this user may open (a container is given a subset of the host's GPUs through the device
    cgroup / the permissions of /dev/dri/renderD*), then narrowed the way the runtime narrows it:
    ROCR_VISIBLE_DEVICES selects among those, HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES among the remaining ones
    (index lists stop at the first index that does not exist; GPU-<uuid> entries count as one device each).
    SPP_KFD_TOPOLOGY points the scan at another tree (tests)."""
    root = os.environ.get("SPP_KFD_TOPOLOGY", "/sys/class/kfd/kfd/topology/nodes")
    real = "SPP_KFD_TOPOLOGY" not in os.environ
    n = 0
    for path in sorted(glob.glob(os.path.join(root, "*", "properties"))):
        props = {}
        try:
            for ln in open(path):
                kv = ln.split()
                if len(kv) == 2:
                    props[kv[0]] = kv[1]
        except OSError:
            continue
        if int(props.get("simd_count", "0") or 0) <= 0:
            continue
        minor = int(props.get("drm_render_minor", "-1") or -1)
        if real and minor >= 0 and not os.access(f"/dev/dri/renderD{minor}", os.R_OK | os.W_OK):
            continue
        n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is None:
            continue
        kept = 0
        for tok in (t.strip() for t in val.split(",")):
            if tok.lstrip("-").isdigit():
                if not 0 <= int(tok) < n:
                    break
                kept += 1
            elif tok:
                kept += 1
            else:
                break
        n = min(n, kept)
    return n


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N rank processes as
    CHILDREN of this one, one per GPU, relay rank 0's JSON line on stdout and return the children's status -- the
    counterpart of the reference's mp.spawn(ddp_main, nprocs=num_devices_per_node) (driver/main.py:337-366).
    This process has not imported torch and never loads the HIP runtime: it runs before `import torch` below and
    counts the GPUs from sysfs (visible_gpu_count).  --launch-dry-run prints what it would do, and whether
    libamdhip64 is mapped into this process, instead of starting anything (tests/test_abi_and_host.py)."""
    import socket
    import subprocess
    n_dev = visible_gpu_count()
    counted_by = "kfd topology (sysfs)"
    if n_dev < a.gpus and "SPP_KFD_TOPOLOGY" not in os.environ:
        # sysfs shows fewer GPUs than asked for (a sandbox may hide the topology): ask a CHILD process that is free to
        # bring up the HIP runtime and exits -- this process still never loads it
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                               text=True, timeout=180)
            n_child = int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
        except Exception:  # noqa: BLE001
            n_child = 0
        if n_child > n_dev:
            n_dev, counted_by = n_child, "child process (torch.cuda.device_count)"
    if a.launch_dry_run:
        maps = open("/proc/self/maps").read()
        print(json.dumps({"launcher": True, "visible_gpus": n_dev, "counted_by": counted_by, "ranks_wanted": a.gpus, "torch_imported": "torch" in sys.modules,
                          "libamdhip64_mapped": "libamdhip64" in maps, "libhsa_runtime_mapped": "libhsa-runtime64" in maps}), flush=True)
    if n_dev < a.gpus:
        print(f"[bench] --gpus {a.gpus} needs {a.gpus} GPUs, this node shows {n_dev}: refusing to run fewer ranks "
              f"than asked for", file=sys.stderr, flush=True)
        return 2
    if a.launch_dry_run:
        return 0
    with socket.socket() as sk:                       # a free rendezvous port on the loop-back interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    argv = [v for v in sys.argv[1:] if v != "--launch-dry-run"]
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # only rank 0 prints the line; whatever else a rank writes to stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"[bench] rank {procs.index(p)} exited with status {code}: stopping the other ranks",
                          file=sys.stderr, flush=True)
                    for q in pending:                 # exactly the processes started above
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# The launcher case is decided HERE, before torch (which maps libamdhip64 into the process) is imported: the parent of the
# rank processes stays free of the GPU runtime.
if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse()
    if _a.gpus > 1 or _a.launch_dry_run:
        raise SystemExit(launch_ranks(_a))



def confine_to_cores(n: int) -> list:
    """--cpu-cores N: the host budget of one rank on an 8-rank node, rehearsed on one GPU.  The process (and every thread
    it starts later: torch's, the Session's launcher / exchanger, RCCL's proxies) is confined to the first N cores of its
    affinity mask.  Runs before `import torch`: nothing has touched the GPU yet.  The reference budgets this explicitly
    (utils/exp_driver.py:48-49 num_workers per trainer, driver/parser.py:87-88)."""
    n = max(1, n)
    r = int(os.environ.get("LOCAL_RANK", "0"))       # every rank its own N cores (wrapping when the mask is short)
    allc = sorted(os.sched_getaffinity(0))
    cores = [allc[(r * n + k) % len(allc)] for k in range(min(n, len(allc)))]
    os.sched_setaffinity(0, cores)
    return cores


if __name__ == "__main__":
    _cores = [v for k, v in zip(sys.argv, sys.argv[1:]) if k == "--cpu-cores"] + \
             [k.split("=", 1)[1] for k in sys.argv if k.startswith("--cpu-cores=")]
    if _cores and int(_cores[-1]) > 0:
        confine_to_cores(int(_cores[-1]))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


class TorchSAGE(torch.nn.Module):
    """Consumer stand-in for the epoch-time figure: the reference's SAGE (driver/models.py:19-56:
    3 x SAGEConv(bias=False, mean aggregation) + ReLU + dropout 0.5 + log_softmax) written with
    plain torch ops on the MFG's CSR, because PyG / torch_sparse are not installed in this image.
    It is NOT part of the product (the model step stays PyG in SALIENT++) and is never in `value`."""

    def __init__(self, in_c, hid, out_c, num_layers=3):
        super().__init__()
        dims = [in_c] + [hid] * (num_layers - 1) + [out_c]
        self.lin_l = torch.nn.ModuleList(torch.nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(num_layers))
        self.lin_r = torch.nn.ModuleList(torch.nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(num_layers))
        self.num_layers = num_layers

    @staticmethod
    def mean_aggregate(x, rowptr, col, n_dst):
        cnt = rowptr[1:] - rowptr[:-1]
        row = torch.repeat_interleave(torch.arange(n_dst, device=x.device), cnt)
        out = torch.zeros((n_dst, x.size(1)), dtype=x.dtype, device=x.device).index_add_(0, row, x[col])
        return out / cnt.clamp(min=1).unsqueeze(-1).to(x.dtype)

    def forward(self, x, adjs):
        x = x.to(torch.float)
        for i, (adj_t, _e_id, size) in enumerate(adjs):
            rowptr, col, _ = adj_t.csr()
            x_target = x[:size[1]]
            x = self.lin_l[i](self.mean_aggregate(x, rowptr, col, size[1])) + self.lin_r[i](x_target)
            if i != self.num_layers - 1:
                x = torch.nn.functional.dropout(torch.relu(x), p=0.5, training=self.training)
        return torch.log_softmax(x, dim=-1)


def make_model_step(F, n_classes, hidden, layers, hip=True, arch="sage", ddp=False):
    """The training step of fast_trainer/train.py:15-71 (forward, nll_loss, backward, Adam) over one PreparedBatch, as a
    closure: the SAME model / optimiser serves the windowed legs and the whole-epoch legs."""
    dev = torch.device("cuda", torch.cuda.current_device())
    if hip:
        from salient_plusplus_amd.models import GAT, SAGE
        model = (GAT if arch == "gat" else SAGE)(F, hidden, n_classes, layers).to(dev)
    else:
        model = TorchSAGE(F, hidden, n_classes, layers).to(dev)
    if ddp:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], broadcast_buffers=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)    # one multi-tensor launch per step

    def step(b):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.nll_loss(model(b.x, b.adjs), b.y.reshape(-1))
        loss.backward()
        opt.step()
    return step


def measure_epochs(make_iter, shuffler, get_idx, n_epochs, first_epoch, step=None, distributed=False):
    """`n_epochs` WHOLE epochs, each timed wall-clock the way the reference times them (fast_trainer/train.py:223-316,
    driver/drivers/base.py:298-423): seed shuffle -> iterator (Session) creation -> every batch [-> model step] -> the
    iterator's StopIteration -> synchronize.  Nothing is multiplied: epoch boundaries, Session set-up, ragged last groups
    and whatever once-per-Session work there is are inside the clock.  The first epoch is kept apart (the reference drops
    it as warm-up).  N > 1: a barrier on both sides, the slowest rank's time."""
    dev = torch.device("cuda", torch.cuda.current_device())
    times, nb = [], 0
    for e in range(n_epochs):
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        shuffler.set_epoch(first_epoch + e)
        it = make_iter(get_idx())
        nb, b = 0, None
        for (b,) in it:
            if step is not None:
                step(b)
            nb += 1
        torch.cuda.synchronize()
        if distributed:
            q = getattr(it, "quiesce", None)
            if q is not None:
                q()
            dist.barrier()
            torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        del it, b
    if distributed and times:
        t = torch.tensor(times, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times = [float(v) for v in t.cpu().tolist()]
    steady = times[1:]
    return {"epochs": n_epochs, "batches": nb, "first_s": times[0] if times else None,
            "steady_s_all": [round(v, 5) for v in steady],
            "steady_s_mean": (sum(steady) / len(steady)) if steady else None,
            "ms_per_batch_steady": (1e3 * sum(steady) / len(steady) / max(1, nb)) if steady else None,
            "clock": "perf_counter around shuffle + iterator creation + all batches + synchronize, per epoch"}


def model_step_timing(feeder, step, steps=64, warm=16, windows=6):
    """ms/step of fwd+bwd+Adam with one resident batch re-used, and with the data path feeding it (the
    training step of fast_trainer/train.py:15-71).  Measured like the data path: `windows` back-to-back
    windows of `steps` steps after `warm` untimed ones, reported: the MEAN over all windows (their total
    time / their total steps) with every window kept in the line -- a fresh process pays a few
    multi-millisecond allocator growths while the model's batch-size-dependent temporaries meet their
    largest shapes, and one 24-step sample after 4 warm-up steps (round 2) carried them into the figure.
    `step`: make_model_step's closure.  Returns (model_only_ms, with_data_ms, detail)."""
    dev = torch.device("cuda", torch.cuda.current_device())

    show = os.environ.get("SPP_BENCH_STEP_TIMES") == "1"

    def run_windows(get, tag, warm_steps):
        for _ in range(warm_steps):
            step(get())
        torch.cuda.synchronize()
        out = []
        for w in range(windows):
            host = []
            t0 = time.perf_counter()
            for _ in range(steps):
                ts = time.perf_counter()
                step(get())
                host.append((time.perf_counter() - ts) * 1e6)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / steps * 1e3)
            if show:
                print(f"[bench] model step ({tag}) window {w}: {out[-1]:.3f} ms/step; host us per step " +
                      " ".join(f"{v:.0f}" for v in host), file=sys.stderr, flush=True)
        return sum(out) / len(out), out

    def n_dev_allocs():
        return int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0))

    fixed = feeder.next()
    a0 = n_dev_allocs()
    m_only, w_only = run_windows(lambda: fixed, "resident batch", warm)
    a1 = n_dev_allocs()
    # the data path has been standing still, every slot full, while the resident-batch leg ran: the first ~50 steps after it
    # resumes (three sampling groups, the refills of all of them) are a transient of this benchmark's leg switch, not of a
    # training loop -- they are warm-up (the first 64-step window used to carry them: 1.21-1.25 against 1.13-1.15 ms)
    warm_data = max(warm, 4 * 16)
    m_data, w_data = run_windows(feeder.next, "with data path", warm_data)
    a2 = n_dev_allocs()
    detail = {"windows": windows, "steps_each": steps, "warmup_steps": warm, "warmup_steps_with_data_path": warm_data,
              "reported": "mean over all windows",
              # (the medians: robust against one allocator stall -- at S-mag on ONE GPU 98 % of the HBM is in use and torch's
              # caching allocator meets a retry, ~1 s, somewhere in the with-data windows)
              "model_only_ms_median": sorted(w_only)[len(w_only) // 2], "with_data_path_ms_median": sorted(w_data)[len(w_data) // 2],
              "model_only_ms_all": [round(v, 4) for v in w_only], "with_data_path_ms_all": [round(v, 4) for v in w_data],
              # hipMalloc calls of torch's caching allocator during each leg (warm-up included)
              "torch_device_allocs": {"resident_batch": a1 - a0, "with_data_path": a2 - a1}}
    return m_only, m_data, detail


def count_edges(batch) -> int:
    return sum(int(adj.adj_t.nnz()) for adj in batch.adjs)


def edges_and_sampler_bytes(batch):
    """(sampled edges, sampler_algorithmic_bytes) in one pass over the hops: this runs inside the timed loop, and the short
    workloads are host bound"""
    edges = alg = 0
    for adj in batch.adjs:
        e = int(adj.adj_t.nnz())
        n_src, n_dst = adj.size
        edges += e
        alg += 24 * int(n_dst) + 16 * e + 8 * (int(n_src) - int(n_dst))
    return edges, alg


def sampler_algorithmic_bytes(batch) -> int:
    """SURVEY 8(d), per hop: T (two rowptr reads + the out_rowptr write per target) x 24 + E (col read + out_col write
    per sampled edge) x 16 + dU (the n_id write per new node) x 8, restating sample_cpu.hpp:25-143."""
    tot = 0
    for adj in batch.adjs:
        n_src, n_dst = int(adj.size[0]), int(adj.size[1])
        tot += 24 * n_dst + 16 * int(adj.adj_t.nnz()) + 8 * (n_src - n_dst)
    return tot


class EpochFeeder:
    """Endless stream of device batches: a new FastSampler iterator (epoch) whenever one runs out,
    with the reference's per-epoch seeded shuffle (fast_trainer/shufflers.py:25-29)."""

    def __init__(self, make_iter, shuffler, get_idx):
        self.make_iter, self.shuffler, self.get_idx = make_iter, shuffler, get_idx
        self.epoch = 0
        self.devit = None
        self._xb = (0, 0)

    def exchange_bytes(self):
        """(sent, received) bytes of the native exchange over all epochs so far (0, 0 without one)."""
        sess = getattr(getattr(self.devit, "it", None), "session", None)
        cur = sess.exchange_bytes() if sess is not None and getattr(sess, "native_exchange", False) else (0, 0)
        return self._xb[0] + cur[0], self._xb[1] + cur[1]

    def _new_epoch(self):
        t0 = time.perf_counter()
        if self.devit is not None:
            self._xb = self.exchange_bytes()
            # finish the old epoch's Session first: its sampler (workspace, exchange buffers) goes back to
            # the pool and the new Session reuses it instead of building a second one
            self.devit = None
        t1 = time.perf_counter()
        self.shuffler.set_epoch(self.epoch)
        idx = self.get_idx()
        t2 = time.perf_counter()
        self.devit = self.make_iter(idx)
        if os.environ.get("SPP_BENCH_STEP_TIMES") == "1":
            print(f"[bench] epoch {self.epoch} set-up: old iterator released {1e3 * (t1 - t0):.2f} ms, seed order "
                  f"{1e3 * (t2 - t1):.2f} ms, new iterator {1e3 * (time.perf_counter() - t2):.2f} ms", file=sys.stderr, flush=True)
        self.epoch += 1

    def quiesce(self):
        q = getattr(self.devit, "quiesce", None)
        if q is not None:
            q()

    def next(self):
        if self.devit is None:
            self._new_epoch()
        while True:
            try:
                return next(self.devit)[0]
            except StopIteration:
                pass
            # outside the except block: while the exception is being handled its traceback keeps the old
            # iterator (and its Session) alive, the pooled sampler is still taken and the new epoch would build
            # a second one (9 ms and 2.4 GB)
            self._new_epoch()


def cpu_baseline(wl_host, sizes, batch_size, seconds, threads):
    """The CPU path timed beside the GPU one on this host's cores, on a bounded sample of the same
    workload: the compiled reference (oracle/_ref) when it is present, else the oracle port."""
    import numpy as np
    rowptr, col, x, y, idx = wl_host
    ref_path = os.path.join(ROOT, "oracle", "_ref", "fast_sampler.so")

    def run_reference(n_batches):
        sys.path.insert(0, os.path.dirname(ref_path))
        import fast_sampler as ref            # the unmodified reference module (built by oracle/build_ref.sh)
        cfg = ref.Config()
        cfg.x_cpu, cfg.x_gpu, cfg.y = x, torch.empty(0), y.unsqueeze(-1)
        cfg.rowptr, cfg.col, cfg.idx = rowptr, col, idx[:n_batches * batch_size].contiguous()
        cfg.batch_size, cfg.sizes = batch_size, list(sizes)
        cfg.skip_nonfull_batch, cfg.pin_memory, cfg.distributed = False, False, False
        cfg.force_exact_num_batches, cfg.exact_num_batches = False, 0
        cfg.count_remote_frequency, cfg.use_cache = False, False
        t0 = time.perf_counter()
        s = ref.Session(threads, 48, cfg)
        edges = 0
        nb = 0
        while True:
            b = s.blocking_get_batch()
            if b is None:
                break
            edges += sum(int(a[1].numel()) for a in b[2])
            nb += 1
        dt = time.perf_counter() - t0
        del s
        return edges, nb, dt

    def run_port(n_batches):
        from oracle import oracle as orc
        ranges = orc.batch_ranges(n_batches * batch_size, batch_size)
        st = orc.epoch_run(rowptr.numpy(), col.numpy(), x.numpy(), y.numpy(), idx.numpy(), ranges, sizes, threads)
        return int(st.sampled_edges), int(st.batches), float(st.seconds)

    kind, run = "port", run_port
    if os.path.exists(ref_path):
        try:
            run_reference(1)
            kind, run = "reference", run_reference
        except Exception as e:  # noqa: BLE001
            print(f"[bench] reference module unusable here ({e}); timing the oracle port", file=sys.stderr)
    probe_batches = max(2, 2 * threads)
    e, nb, dt = run(probe_batches)
    per_batch = dt / max(nb, 1)
    avail = max(1, idx.numel() // batch_size)
    want = max(probe_batches, int(seconds / max(per_batch, 1e-6)))
    e = nb = 0
    dt = 0.0
    while nb < want:                      # whole passes over the epoch's seeds until ~`seconds` of CPU work
        e1, nb1, dt1 = run(min(avail, want - nb))
        e, nb, dt = e + e1, nb + nb1, dt + dt1
    return {"value": e / dt, "unit": "sampled-edges/s", "cores": threads, "kind": kind,
            "sample": f"{nb} batches of {batch_size} seeds ({e} sampled edges) in {dt:.2f}s, "
                      f"{threads} worker threads, incl. feature/label slicing",
            "batches_per_s": nb / dt}


def _trace(msg):
    """progress markers on stderr (SPP_BENCH_TRACE=1): where a multi-rank run stops making progress"""
    if os.environ.get("SPP_BENCH_TRACE") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')}] {time.strftime('%H:%M:%S')} {msg}", file=sys.stderr, flush=True)


def main():
    a = parse()
    if os.environ.get("SPP_BENCH_TRACE") == "1":
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("SPP_BENCH_TRACE_AFTER", "60")), exit=False)
    if (a.gpus > 1 or a.launch_dry_run) and "WORLD_SIZE" not in os.environ:
        raise SystemExit("bench.py: the launcher case is handled before torch is imported (run the file as a script)")
    # stdout carries ONE JSON line and nothing else: libraries that announce themselves there (RCCL prints its version
    # banner to stdout when a communicator is created) are sent to stderr, the line is written to the saved descriptor
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks: the line would "
                         f"not describe the run that was asked for")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X data path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or a.force_distributed
    if a.slots <= 0:
        a.slots = int(os.environ.get("SPP_MAX_SLOTS", "64"))
    if a.prime < 0:
        a.prime = 3 * a.slots
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from salient_plusplus_amd import _native as nat
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
    from salient_plusplus_amd.fast_trainer.shufflers import DistributedShuffler, FederatedDistributedShuffler, Shuffler
    from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher, DevicePrefetcher
    from salient_plusplus_amd.synthetic import make_workload

    L = nat.load()
    nat.require_device()
    _trace("process group up, building the workload")
    t_build = time.perf_counter()
    wl = make_workload(a.workload, seed=1234, device=dev)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    N, F = wl.num_nodes, wl.x.size(1)
    sizes, bs = wl.fanouts, wl.batch_size
    if a.fanouts:
        sizes = [int(v) for v in a.fanouts.split(",")]
    y2 = wl.y.unsqueeze(-1)

    if not distributed:
        n_train = wl.train_idx.numel()
        cfg = FastSamplerConfig(
            x_cpu=wl.x, x_gpu=torch.empty(0), y=y2, rowptr=wl.rowptr, col=wl.col,
            idx=wl.train_idx, batch_size=bs, sizes=sizes, skip_nonfull_batch=False, pin_memory=False,
            distributed=False, partition_book=None, cache=fs.Cache(), force_exact_num_batches=True,
            exact_num_batches=max(1, n_train // bs), count_remote_frequency=False, use_cache=False)
        sampler = FastSampler(4, a.slots, cfg)
        shuffler = Shuffler(wl.train_idx)

        def make_iter(idx):
            sampler.idx = idx
            return DevicePrefetcher([dev], iter(sampler))
        feeder = EpochFeeder(make_iter, shuffler, shuffler.get_idx)
        parallelism = "single"
        x_local = wl.x
    else:
        # contiguous range partition of the features; topology and labels replicated (dataset.py:205-211)
        offsets = torch.linspace(0, N, world + 1).long()
        offsets[-1] = N
        lo, hi = int(offsets[rank]), int(offsets[rank + 1])
        x_local = wl.x[lo:hi].contiguous()
        pb = fs.RangePartitionBook(rank, world, offsets)
        _trace("workload built, building the feature cache")
        # VIP cache (ddp.py:417-570): the remote vertices most likely to be touched by this rank's
        # mini-batches (analytic model, ddp.py:135-239, on the GPU), alpha * N / P rows fetched from
        # their owners once.  --cache-strategy degree-desc keeps the earlier top-degree proxy.
        from salient_plusplus_amd.fast_trainer.vip_cache import fetch_cache_rows
        # the vertices this rank's mini-batches start from (ddp.py:33-34 feeds the VIP model with
        # split_idx_parts[rank]['train'])
        vip_seeds = wl.train_idx[(wl.train_idx >= lo) & (wl.train_idx < hi)].contiguous() \
            if a.seed_scheme == "federated" else wl.train_idx
        n_cache = int(a.cache_frac * N / world) if world > 1 else 0
        if n_cache > 0 and a.cache_strategy == "degree-desc":
            deg_remote = (wl.rowptr[1:] - wl.rowptr[:-1]).clone()
            deg_remote[lo:hi] = -1
            wanted = torch.topk(deg_remote, n_cache).indices.sort().values
            cv, cf = fetch_cache_rows(pb, wanted, x_local)
            cache = fs.Cache(rank, world, cv, cf)
        elif n_cache > 0:
            # create_vip_cache (ddp.py:417-570) in its two halves: the ranking and the collective fetch of the
            # rows from their owners
            from salient_plusplus_amd.fast_trainer.vip_cache import rank_remote_vertices
            wanted = rank_remote_vertices(a.cache_strategy, pb, N, int(N / world * a.cache_frac), rowptr=wl.rowptr,
                                          col=wl.col, train_idx=vip_seeds, fanouts=sizes, batch_size=bs)
            cv, cf = fetch_cache_rows(pb, wanted, x_local)
            cache = fs.Cache(rank, world, cv, cf)
            n_cache = int(cache.cached_vertices.numel())
        else:
            cache = fs.Cache()
        _trace(f"cache built ({n_cache} rows)")
        # This benchmark issues no collectives of its own inside the timed loop, so the exchanges may be
        # issued by the session thread as soon as a group is sampled (most overlap); a training loop with
        # DDP all-reduces keeps the library default (SPP_EXCHANGE_ISSUE=consumer).
        os.environ.setdefault("SPP_EXCHANGE_ISSUE", "thread")
        # a missing peer / mismatched batch sequence ends in a diagnostic and a non-zero exit well inside the
        # driver's time limit instead of a kill at the limit (the library default is 300 s)
        os.environ.setdefault("SPP_EXCHANGE_TIMEOUT_S", "120")
        # collective: every rank joins the RCCL communicator of the native exchange; if any rank cannot,
        # all of them fall back to the torch.distributed transport together
        try:
            native = fs.native_comm() is not None
            native_err = None
        except Exception as e:  # noqa: BLE001
            native, native_err = False, repr(e)
        flag = torch.tensor([1 if native else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not int(flag.item()):
            if native_err and rank == 0:
                print(f"[bench] native exchange unavailable: {native_err}", file=sys.stderr, flush=True)
            os.environ["SPP_DIST_TRANSPORT"] = "torch"
            native = False
        # Seeds.  "federated" (default; the reference launcher's default, utils/exp_driver.py:113 and
        # shufflers.py:92-101): every rank shuffles the training vertices of ITS partition, which is what
        # lets a locality-preserving partition keep most sampled neighbours local.  "global": one seeded
        # permutation of all training ids, contiguous 1/N slice per rank (shufflers.py:32-45).  Every
        # rank runs the same number of batches of `bs` seeds (force_exact_num_batches keeps the
        # exchanges aligned): the smallest pool decides.
        if a.seed_scheme == "federated":
            mine = wl.train_idx[(wl.train_idx >= lo) & (wl.train_idx < hi)].contiguous()
            shuffler = FederatedDistributedShuffler(mine)
            get_idx = shuffler.get_idx
            pool = torch.tensor([mine.numel()], device=dev)
            if world > 1:
                dist.all_reduce(pool, op=dist.ReduceOp.MIN)
            n_local = int(pool.item())
        else:
            shuffler = DistributedShuffler(wl.train_idx, world)
            get_idx = lambda: shuffler.get_idx(rank)                       # noqa: E731
            n_local = wl.train_idx.numel() // world
        cfg = FastSamplerConfig(
            x_cpu=torch.empty((0, F), dtype=wl.x.dtype), x_gpu=x_local, y=y2, rowptr=wl.rowptr, col=wl.col,
            idx=get_idx(), batch_size=bs, sizes=sizes, skip_nonfull_batch=False, pin_memory=False,
            distributed=True, partition_book=pb, cache=cache, force_exact_num_batches=True,
            exact_num_batches=max(1, n_local // bs), count_remote_frequency=False, use_cache=n_cache > 0)
        sampler = FastSampler(4, a.slots, cfg)

        # Parity on the real transport, outside the timed region: every rank holds the whole synthetic
        # feature matrix, so one group of batches through the native exchange can be compared bit for
        # bit with x_full[n_id] (what a single GPU would deliver).  Collective: all ranks run it.
        exchange_verified = None
        verified_per_rank = None
        rccl_world = int(L.spp_comm_world(fs.native_comm().handle)) if native else 0
        if world > 1 and native and rccl_world != world:
            raise SystemExit(f"bench.py --gpus {a.gpus}: the native exchange communicator spans {rccl_world} ranks, "
                             f"the process group {world}")
        if native and not a.no_verify:
            import dataclasses
            n_check = min(8, max(1, n_local // bs))
            vcfg = dataclasses.replace(cfg, idx=get_idx()[:n_check * bs].contiguous(), exact_num_batches=n_check)
            vit = iter(FastSampler(4, a.slots, vcfg))
            good = vit.session.native_exchange
            for proto in vit:
                good = good and proto.x is not None and bool(torch.equal(proto.x, wl.x[proto.n_id]))
            vit.session.quiesce()
            vit.session.close()
            del vit
            flags = torch.zeros(world, dtype=torch.int32, device=dev)
            flags[rank] = 1 if good else 0
            dist.all_reduce(flags, op=dist.ReduceOp.SUM)
            verified_per_rank = [bool(v) for v in flags.cpu().tolist()]
            exchange_verified = all(verified_per_rank)
            _trace(f"exchange verified: {exchange_verified}")

        def make_iter(idx):
            sampler.idx = idx
            return DeviceDistributedPrefetcher([dev], iter(sampler), pipeline_on=True)
        feeder = EpochFeeder(make_iter, shuffler, get_idx)
        parallelism = f"dp{world}: features range-partitioned {world}-way, {a.cache_strategy} cache " \
                      f"{a.cache_frac:.0%} of N/P rows ({n_cache}), {a.seed_scheme} seeds, {max(1, n_local // bs)} batches per rank and epoch, " + \
                      (f"native RCCL exchange per sampling group (up to 16 batches; all-gather counts, grouped send/recv ids+rows; issued by the "
                       f"{os.environ.get('SPP_EXCHANGE_ISSUE')})"
                       if native else "torch.distributed all_to_all_single, one exchange per sampling group (up to 16 batches)")

    # ---- set-up: first-touch costs (allocator segments, workspace, exchange buffers) ----
    _trace("iterator ready, priming")
    for k in range(max(0, a.prime)):
        feeder.next()
        _trace(f"primed batch {k}")
    gc.collect()
    gc.freeze()           # the long-lived set-up objects need not be re-scanned by a collection inside the timed loop
    # ---- warmup ----
    for _ in range(a.warmup):
        feeder.next()
    torch.cuda.synchronize()
    _trace("warm-up done")
    if distributed:
        feeder.quiesce()      # nothing of the exchange's communicator in flight while the barrier's kernels run
        _trace("quiesced")
        dist.barrier()
        _trace("barrier passed, timing")
    torch.cuda.synchronize()
    # live HIP-event timing of the delivery launches: every launch with group delivery (one per sampling group), every 8th
    # with per-batch delivery (the two timing events around EVERY ~100 us launch cost the delivery queue ~7 us each time)
    prof_every = 1 if os.environ.get("SPP_GROUP_DELIVERY", "0") != "0" else 8
    L.spp_profile_enable(0 if os.environ.get("SPP_BENCH_NO_PROF") == "1" else prof_every)
    # R windows of EXACTLY K steps each, every one bracketed by barrier + synchronize on both sides (the
    # closing bracket of a window is the opening bracket of the next, so the sampler's slots stay full
    # in between).  A single 20-step window is ~3 ms: its closing synchronize also waits for the refill
    # chains the sampler has in flight, which makes one short window noisy.  Reported: the mean over the windows (see
    # below for the one guard); every window is in the line.
    R = a.windows if a.windows > 0 else max(6, 2 * -(-128 // max(1, a.steps)))
    xb0 = feeder.exchange_bytes()
    dev_allocs0 = int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0))
    if os.environ.get("SPP_BENCH_STALL_DUMP"):        # diagnostic: Python stacks of all threads every N seconds of the timed region
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SPP_BENCH_STALL_DUMP"]), repeat=True)
    win = []                                      # (seconds, edges, nodes) per window
    prof_cum = []                                 # cumulative (ms, launches) of the timed delivery launches after each window
    chain_alg_bytes = 0                           # the sampler's algorithmic bytes over the timed region (this rank)
    groups_per_window = []                        # sampling groups opened in each window (a window that opens two carries two chains' work)
    for _w in range(R):
        edges = nodes = 0
        g0 = fs.groups_opened()
        t0 = time.perf_counter()
        step_t = [] if os.environ.get("SPP_BENCH_STEP_TIMES") == "1" else None
        for _ in range(a.steps):
            ts = time.perf_counter()
            b = feeder.next()
            e_, a_ = edges_and_sampler_bytes(b)
            edges += e_
            nodes += b.x.size(0)
            chain_alg_bytes += a_
            if step_t is not None:
                step_t.append((time.perf_counter() - ts) * 1e6)
                if step_t[-1] > 50000:            # a stall: what the caching allocator did meanwhile
                    ms_ = torch.cuda.memory_stats(dev)
                    print(f"[bench] step of {step_t[-1] / 1e3:.0f} ms in window {_w}: torch device allocs {ms_.get('num_device_alloc')} "
                          f"frees {ms_.get('num_device_free')} retries {ms_.get('num_alloc_retries')} reserved "
                          f"{torch.cuda.memory_reserved(dev) / 2**30:.1f} GB, free HBM {torch.cuda.mem_get_info(dev)[0] / 2**30:.1f} GB",
                          file=sys.stderr, flush=True)
        ts = time.perf_counter()
        if os.environ.get("SPP_BENCH_TAIL") == "1":   # diagnostic: how long the closing synchronize waits for the DELIVERIES
            side = getattr(getattr(feeder.devit, "side", None), "stream", None)     # and how long for the chains behind them
            if side is not None:
                side.synchronize()
            torch.cuda.current_stream().synchronize()
            t_del = time.perf_counter()
            torch.cuda.synchronize()
            print(f"[bench] window {_w}: issue {1e6 * (ts - t0):.0f} us, deliveries done +{1e6 * (t_del - ts):.0f} us, "
                  f"everything else (chains in flight) +{1e6 * (time.perf_counter() - t_del):.0f} us", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        if step_t is not None:                    # diagnostic: host time of every step of a window, and of its closing synchronize
            shown = step_t if len(step_t) <= 24 else sorted(step_t)[-8:]
            print(f"[bench] window {_w}: host us per step ({'all' if len(step_t) <= 24 else 'the 8 longest'}) " +
                  " ".join(f"{v:.0f}" for v in shown) + f" | synchronize {(time.perf_counter() - ts) * 1e6:.0f}",
                  file=sys.stderr, flush=True)
        if distributed:
            feeder.quiesce()
            dist.barrier()
        torch.cuda.synchronize()
        win.append((time.perf_counter() - t0, float(edges), float(nodes)))
        groups_per_window.append(fs.groups_opened() - g0)
        # (outside the window's clock) the delivery launches timed so far: per-window in-situ duration of the dominant kernel,
        # which tells a slow window of the GPU (longer launches) from one of the host or the queue (same launches, more gaps)
        _ms, _n, _u = C.c_double(0), C.c_int64(0), C.c_int64(0)
        nat.check(L.spp_profile_read(1 if (distributed and not native) else 0, C.byref(_ms), C.byref(_n), C.byref(_u)))
        prof_cum.append((_ms.value, _n.value))
    xb1 = feeder.exchange_bytes()
    dev_allocs_timed = int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0)) - dev_allocs0
    if os.environ.get("SPP_BENCH_STALL_DUMP"):
        faulthandler.cancel_dump_traceback_later()
    # gather-kernel time, measured live with HIP events on the launching stream
    ms, n_launch, rows = C.c_double(0), C.c_int64(0), C.c_int64(0)
    # SPP_PROF_GATHER: the fused delivery launch (also assembles x with the native exchange);
    # SPP_PROF_ASSEMBLE: the stand-alone assembly of the torch.distributed transport
    prof_kind = 1 if (distributed and not native) else 0
    nat.check(L.spp_profile_read(prof_kind, C.byref(ms), C.byref(n_launch), C.byref(rows)))
    # the sampling chains: HIP events around every chain of a group of batches on its sampling stream (SPP_PROF_CHAIN)
    ch_ms, ch_n, ch_batches = C.c_double(0), C.c_int64(0), C.c_int64(0)
    nat.check(L.spp_profile_read(2, C.byref(ch_ms), C.byref(ch_n), C.byref(ch_batches)))
    L.spp_profile_enable(0)

    stats = torch.tensor(win, dtype=torch.float64, device=dev)      # [R, 3]
    if distributed:
        tmax = stats[:, 0].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)                  # per window: the slowest rank's time
        tot = stats[:, 1:].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)                   # whole-job edges / nodes
        stats = torch.cat([tmax.unsqueeze(1), tot], dim=1)
    win = stats.cpu().tolist()
    # reported: the PLAIN mean over the windows -- the statistic that multiplies back to the timed region
    # (ms_per_step x steps x windows = timed_region_s).  With K not a multiple of the sampler's group the windows differ by
    # where their closing synchronize catches the refill chains, so the median would leave the long ones out; a trimmed
    # mean (without the slowest and the fastest of >= 8 windows; the headline of round 3) is robust against a single stall
    # but is not what the clock around the run sees.  Median and trimmed mean stay in the line as extra keys.
    order = sorted(range(R), key=lambda k: win[k][0])
    kept = order[1:-1] if R >= 8 else order
    dt_trimmed = sum(win[k][0] for k in kept) / len(kept)
    dt, edges, nodes = (sum(w[j] for w in win) / R for j in range(3))
    dt_mean = dt
    srt = [win[k][0] for k in order]
    dt_median = srt[R // 2] if R % 2 else 0.5 * (srt[R // 2 - 1] + srt[R // 2])
    window_ms = [w[0] / a.steps * 1e3 for w in win]
    timed_total_s = sum(w[0] for w in win)

    # ---- the metric's second half: epoch time with the model step consuming the batches -------------
    # N == 1: models.py SAGE / GAT on the single-GPU iterator.  N > 1: the same model under
    # DistributedDataParallel on torch's NCCL process group (driver/drivers/ddp.py:349-350,
    # fast_trainer/train.py:15-71): gradient all-reduces of one communicator interleave with the feature
    # exchanges of another, so the Sessions of this leg issue their exchanges from the consumer thread, at the
    # same program point on every rank (SPP_EXCHANGE_ISSUE=consumer).  All ranks take part.
    # ---- the line, as a function: called at the end, or -- N > 1 -- by the watchdog below when an optional leg hangs ----
    epoch_measured = model_out = p2p_out = None
    sinfo = {}
    legs_state = {"partial": False}

    def emit_line():
        from salient_plusplus_amd.synthetic import LOCALITY
        locality_note = ""
        if a.workload in LOCALITY:
            locality_note = f" (planted {LOCALITY[a.workload][0]}-block locality, {LOCALITY[a.workload][1]:.0%} intra-block edges)"
        if rank == 0:
            # dominant HBM kernel: the feature-row gather (x rows dominate: y rows are 8 B each)
            row_bytes = F * 2
            # read row + write row + the index (int64 at the boundary); the assembly of the partitioned path
            # reads an 8-byte {bucket, row} record and the row's own address instead: + 12
            alg_bytes_per_row = 2 * row_bytes + (12 if distributed else 8)
            # One profiled launch per batch.  N == 1: the fused delivery kernel (x-row gather + the small
            # label gather and int64 widening of the MFG, which are charged to the gather's time but not
            # to its bytes: conservative).  N > 1: the fused assembly kernel.
            x_rows = rows.value
            x_ms = ms.value
            launches_x = n_launch.value
            achieved = (x_rows * alg_bytes_per_row) / (x_ms * 1e-3) / 1e9 if x_ms > 0 else 0.0
            roof = {"bound": "hbm", "kernel": "k_assemble" if prof_kind == 1 else
                    ("k_deliver (assemble from local/received/cache rows)" if distributed else "k_deliver (gather_rows_body)"),
                    "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "avg_launch_ms": x_ms / max(1, launches_x), "launches": launches_x,
                    "algorithmic_bytes_per_row": alg_bytes_per_row, "rows_per_launch": x_rows / max(1, launches_x)}
            # HBM traffic of the same kernel from the PMC passes kept under profiles/ (FETCH_SIZE and
            # WRITE_SIZE in separate rocprofv3 runs, corrected on a known-bytes launch of this access width)
            # (the partitioned path's delivery -- assembly from {local, received, cache} rows -- has a pass of its own)
            import glob
            prof_dir = os.path.join(ROOT, "profiles")

            def newest_first(pattern):
                return tuple(os.path.basename(f) for f in sorted(glob.glob(os.path.join(prof_dir, pattern)), reverse=True))
            pmc_files = newest_first("r0?_deliver_partitioned_pmc.json") if distributed else \
                newest_first("r0?_deliver_pmc_*.json") + newest_first("r0?_gather_pmc_papers.json") + newest_first("r0?_gather_pmc.json")
            for pmc_name in pmc_files:
                pmc_path = os.path.join(ROOT, "profiles", pmc_name)
                if (distributed and not native) or not os.path.exists(pmc_path):
                    continue
                pmc = json.load(open(pmc_path))
                if pmc["shape"]["row_bytes"] == row_bytes:
                    roof["traffic"] = pmc["traffic_bytes_per_row"] * roof["rows_per_launch"]
                    roof["traffic_source"] = (f"profiles/{pmc_name}: PMC bytes/row of this kernel from an earlier rocprofv3 "
                                              f"--pmc pass (NOT measured in this run) x this run's rows/launch")
                    roof["algorithmic_bytes_per_launch"] = alg_bytes_per_row * roof["rows_per_launch"]
                    break
            # The sampler (roofline per kernel family).  Algorithmic bytes of the batches this rank consumed in the
            # timed region / their number = per batch; time = the in-situ span of a chain (all hops of a group of batches, first
            # launch to last on its sampling stream, beside the delivery kernel and the chain of the other sampling stream)
            # / the batches of the group.  traffic: per batch, from the last committed PMC passes over this command.
            roof_s = None
            if ch_n.value > 0 and ch_ms.value > 0:
                alg_pb = chain_alg_bytes / max(1, a.steps * R)
                span_pb = ch_ms.value / max(1, ch_batches.value)                 # ms of chain span per batch
                ach = alg_pb / (span_pb * 1e-3) / 1e9
                roof_s = {"bound": "hbm", "kernel": "sampling chain (k_seed_init, k_hop_count/pick, k_bucket_*, k_hop_flag/rows; all hops)",
                          "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                          "algorithmic_bytes_per_batch": alg_pb, "chain_span_ms_per_batch": span_pb,
                          "avg_chain_span_ms": ch_ms.value / ch_n.value, "batches_per_chain": ch_batches.value / ch_n.value,
                          "chains_timed": ch_n.value,
                          "note": "in-situ span on the chain's own stream: the chains of two slot-sets run concurrently on two sampling "
                                  "streams beside the delivery kernel, so spans overlap (the per-batch cost in the step is smaller than the span)"}
                pmc_txt = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0?_partitioned_pmc_traffic_per_kernel.txt" if distributed
                                                        else "r0?_pipeline_pmc_traffic_per_kernel.txt")))
                # (the committed passes are of S-papers with its own fan-outs: any other workload keeps "traffic": null)
                if pmc_txt and a.workload == "S-papers" and not a.fanouts:
                    mb = 0.0
                    for ln in open(pmc_txt[-1]):
                        f_ = ln.split()
                        if ln.startswith("spp::") and "k_deliver" not in ln and len(f_) >= 4:
                            mb += float(f_[-3]) + float(f_[-2])
                    if mb > 0:
                        roof_s["traffic"] = mb * 1e6
                        roof_s["traffic_source"] = (f"profiles/{os.path.basename(pmc_txt[-1])}: fetch + write MB per batch of the chain's kernels from "
                                                    f"earlier rocprofv3 --pmc passes over this command (NOT measured in this run)")
            if roof_s is not None and not distributed:
                # context for the in-situ span: the chain ALONE on the GPU (two sampling streams, no deliveries), from the
                # committed sampling-only run of this workload's default configuration -- not measured in this run
                lone = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0?_chain_only_trace_report.txt")))
                if lone and a.workload == "S-papers" and not a.fanouts:
                    import re
                    m_ = re.search(r"chain only: .*?([0-9.]+) us/batch", open(lone[-1]).read())
                    if m_:
                        us = float(m_.group(1))
                        roof_s["lone_chain"] = {"us_per_batch": us, "achieved_GBps_algorithmic": roof_s["algorithmic_bytes_per_batch"] / us / 1e3,
                                                "achieved_GBps_traffic": (roof_s["traffic"] / us / 1e3) if roof_s["traffic"] else None,
                                                "source": f"profiles/{os.path.basename(lone[-1])} (tools/microbench.py chain; NOT measured in this run)"}
            # the step as a whole against the memory system: the HBM traffic of one batch (delivery + chain, from the committed PMC
            # passes over this command) over the measured step
            roof_p = None
            if roof.get("traffic") and roof_s is not None and roof_s.get("traffic") and not distributed:
                tot = roof["traffic"] + roof_s["traffic"]
                ach = tot / (dt / a.steps) / 1e9
                roof_p = {"bound": "hbm", "what": "the whole step: counter traffic of one batch's delivery + sampling chain over ms_per_step",
                          "traffic_bytes_per_batch": tot, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                          "traffic_source": "profiles/ PMC passes (NOT measured in this run) x this run's step"}
            out = {
                "metric": "sampled_edges_per_sec", "value": edges / dt, "unit": "sampled-edges/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64/fp16-copy",
                "data": "synthetic",
                "config": {"workload": f"{a.workload}{locality_note}: N={N} nnz={int(wl.col.numel())} F={F} fp16, "
                                       f"fanout {sizes}, batch {bs}, all features in HBM",
                           "parallelism": parallelism, "slots_in_flight": a.slots,
                           # which form of the sampling chain the library chose by itself (spp_sampler_get_info)
                           "sampler_variant": {k_: sinfo.get(k_) for k_ in ("col32", "deg_tags", "row_stubs", "rng_arena", "idbits", "tag_cap",
                                                                            "generic", "fused_pick", "flag_tiled", "rows_coalesced",
                                                                            "bucket_log2", "dedup_table_slots")} if sinfo else None},
                "windows": {"n": R, "steps_each": a.steps, "reported": "mean over all windows (timed_region_s / all steps)",
                            "ms_per_step_min": min(window_ms), "ms_per_step_median": dt_median / a.steps * 1e3,
                            "ms_per_step_mean": dt_mean / a.steps * 1e3,
                            "ms_per_step_trimmed_mean": dt_trimmed / a.steps * 1e3,   # without the slowest and the fastest of >= 8 windows
                            "ms_per_step_max": max(window_ms), "timed_region_s": timed_total_s,
                            "ms_per_step_all": [round(v, 5) for v in window_ms],
                            # sampling groups (one chain of up to 16 batches each) rank 0 opened in each window, and the windows
                            # that opened two or more: K steps are K / 16 groups, so with K = 20 one window in four carries two
                            # chains' work (a structural long window, not a stall)
                            "groups_opened_all": groups_per_window,
                            "opens_two_groups": [k for k, n_ in enumerate(groups_per_window) if n_ >= 2],
                            # in-situ duration of the timed delivery launches of each window (rank 0; live HIP events)
                            "deliver_us_all": [round(1e3 * (b[0] - a_[0]) / max(1, b[1] - a_[1]), 1)
                                               for a_, b in zip([(0.0, 0)] + prof_cum[:-1], prof_cum)]},
                "timed_region_s": timed_total_s,          # all R windows (also under "windows")
            "optional_legs_timed_out": legs_state["partial"],   # N > 1 only: the watchdog printed this line
                "host": {"cpu_cores_flag": a.cpu_cores, "affinity_cores": len(os.sched_getaffinity(0)), "cpu_share": host_cpu_share()},
                "hbm": None if legs_state["partial"] else
                       {"free_gb": round(torch.cuda.mem_get_info(dev)[0] / 2**30, 2), "total_gb": round(torch.cuda.mem_get_info(dev)[1] / 2**30, 2),
                        "torch_reserved_gb": round(torch.cuda.memory_reserved(dev) / 2**30, 2),
                        "torch_alloc_retries": int(torch.cuda.memory_stats(dev).get("num_alloc_retries", 0)),
                        "torch_device_allocs": int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0)),
                        "torch_device_allocs_in_timed_region": dev_allocs_timed},   # hipMalloc calls of the caching allocator
                "priming_steps": max(0, a.prime),
                "batches_per_s": a.steps * world / dt,
                # (an EXTRAPOLATION: batches x ms/step of the windows; the measured epochs are under "epoch_measured")
                "epoch_time_s_data_path_only": (wl.train_idx.numel() // bs) / (a.steps / dt) if not distributed else None,
                "mfg_nodes_per_batch": nodes / (a.steps * world), "sampled_edges_per_batch": edges / (a.steps * world),
                "graph_build_s": t_build,
                # once-per-process work the timed windows lean on and never pay: the int32 neighbour array, the row stubs (both
                # once per graph) and the mt19937 streams of the whole epoch (once per range table; the pooled sampler keeps
                # them across epochs).  The measured epochs' FIRST epoch is where a training run would see them.
                "setup": {"col32_ms": sinfo.get("col32_ms"), "col32_GB": (sinfo.get("col32_bytes") or 0) / 1e9,
                          "row_stubs_ms": sinfo.get("row_stubs_ms"), "row_stubs_GB": (sinfo.get("row_stubs_bytes") or 0) / 1e9,
                          "rng_arena_ms": sinfo.get("rng_arena_ms"), "rng_arena_GB": (sinfo.get("rng_arena_bytes") or 0) / 1e9,
                          "rng_arena_batches": sinfo.get("rng_arena_batches"),
                          "rng_arena_ms_per_batch_amortised_over_one_epoch":
                              (sinfo.get("rng_arena_ms") or 0.0) / max(1, sinfo.get("rng_arena_batches") or 1),
                          "charged_to": "neither `value` nor the windows: set-up (first Session of the process)"} if sinfo else None,
                "epoch_measured": epoch_measured,
                "roofline": roof,
                "roofline_sampler": roof_s,
                "roofline_pipeline": roof_p,
            }
            if distributed and native:
                # rank 0's share of the exchange over the timed region (the exchange runs ahead of the consumer
                # by up to the slot-sets in flight, so this is within one group of the bytes of the K batches)
                sent, recv = xb1[0] - xb0[0], xb1[1] - xb0[1]
                n_timed = a.steps * R
                out["exchange"] = {"transport": "RCCL grouped send/recv over xGMI",
                                   "rccl_world": rccl_world,          # ranks the native communicator really spans
                                   "verified_bit_exact_vs_full_table": exchange_verified,
                                   "verified_per_rank": verified_per_rank,
                                   "timeout_s": float(os.environ.get("SPP_EXCHANGE_TIMEOUT_S", "300")),
                                   "rank0_sent_MB_per_batch": sent / n_timed / 1e6,
                                   "rank0_received_MB_per_batch": recv / n_timed / 1e6,
                                   "rank0_GBps_out": sent / timed_total_s / 1e9, "rank0_GBps_in": recv / timed_total_s / 1e9,
                                   "xgmi_peak_GBps_per_gpu": 7 * 153.0}
            if p2p_out is not None:
                out["exchange_p2p"] = p2p_out
            if model_out is not None:
                out.update(model_out)
            if not a.no_cpu_baseline and not distributed:
                threads = host_cpu_share()
                host = (wl.rowptr.cpu(), wl.col.cpu(), wl.x.cpu(), wl.y.cpu(), shuffler.get_idx().cpu())
                out["cpu_baseline"] = cpu_baseline(host, sizes, bs, a.cpu_seconds, threads)
                out["cpu_baseline"]["cores_source"] = (f"min(sched_getaffinity = {len(os.sched_getaffinity(0))}, cgroup cpu.max quota) "
                                                       f"= {threads}; os.cpu_count() shows {os.cpu_count()}")
                out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            sys.stdout.flush()
            os.write(line_fd, (json.dumps(out) + "\n").encode())

    # N > 1: the legs below are optional extras behind the measured windows, and every one of them is a sequence of
    # collectives -- a rank that falls out of step would hang the others, and the line (printed last) with it.  A watchdog
    # prints the line with what has been measured so far (`optional_legs_timed_out`) and ends the process.
    watchdog = None
    if distributed:
        import threading

        def _deadline():
            try:
                legs_state["partial"] = True
                if rank == 0:
                    emit_line()
            finally:
                os._exit(0)
        watchdog = threading.Timer(float(os.environ.get("SPP_BENCH_LEG_TIMEOUT_S", "420")), _deadline)
        watchdog.daemon = True
        watchdog.start()
    model_out = None
    layers = a.layers if a.layers > 0 else len(sizes)
    n_classes = 47
    model_name = f"{a.model.upper()} {layers}x{a.hidden}"
    nb_epoch = (wl.train_idx.numel() // bs) if not distributed else max(1, n_local // bs)
    # ---- whole epochs, wall-clock (no multiplication): leg (a), the data path alone ----
    epoch_measured = None
    next_epoch = [feeder.epoch + 1]

    def run_epochs(mk, step=None):
        r = measure_epochs(mk, shuffler, get_idx if distributed else shuffler.get_idx, a.epochs, next_epoch[0], step=step,
                           distributed=distributed)
        next_epoch[0] += a.epochs
        return r
    # which chain variant ran and what its one-off tables cost (read before the timed Session goes away)
    sess0 = getattr(getattr(feeder.devit, "it", None), "session", None)
    sinfo = sess0.sampler_info() if sess0 is not None else {}
    del sess0        # (a live reference would keep the Session, and with it the pooled sampler, away from the next one)
    if a.epochs > 0:
        if distributed:
            feeder.quiesce()
        feeder.devit = None                      # the windows' Session ends; its pooled sampler serves the epochs' Sessions
        gc.collect()
        _trace("measuring whole epochs (data path alone)")
        epoch_measured = {"data_path_only": run_epochs(make_iter)}
    if not a.no_model_step:
        try:
            if distributed:
                feeder.quiesce()
                feeder.devit = None                  # the thread-issued Session ends here
                gc.collect()
                os.environ["SPP_EXCHANGE_ISSUE"] = "consumer"
                dist.barrier()
            step = make_model_step(F, n_classes, a.hidden, layers, hip=True, arch=a.model, ddp=distributed)
            m_only, m_data, m_detail = model_step_timing(feeder, step)
            if distributed:
                feeder.quiesce()
                t = torch.tensor([m_only, m_data], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)             # the slowest rank's step
                m_only, m_data = (float(v) for v in t.cpu().tolist())
            model_out = {"epoch_time_s_with_model_step": nb_epoch * m_data / 1e3,
                         "model_step": {"model": f"{model_name} (models.py: HIP message passing + library GEMMs, "
                                                 f"fp32, fused ReLU+dropout, Adam(fused=True); one layer per hop of {sizes})" +
                                                 (f", DistributedDataParallel over {world} ranks, exchanges issued by the consumer"
                                                  if distributed else ""),
                                        "hidden": a.hidden, "layers": layers,
                                        "batches_per_epoch_and_rank": nb_epoch,
                                        "ms_per_step_model_only_resident_batch": m_only,
                                        "ms_per_step_with_data_path": m_data, "timing": m_detail}}
            if epoch_measured is not None:
                # leg (b): the default consumer -- x delivered, the model step on every batch
                if distributed:
                    feeder.quiesce()
                feeder.devit = None
                gc.collect()
                _trace("measuring whole epochs (with the model step)")
                epoch_measured["with_model_step"] = run_epochs(make_iter, step)
            if distributed and native and a.model == "sage" and not a.no_fused_leg:
                # Row g1 on the partitioned path: Session(row_refs) delivers MFG + labels + where every row lives (local
                # partition / cache / the rows received for the batch) and models.SAGE's first layer reads from there
                # (spp_sage_operand_forward_rows; bit-identical operand, tests/test_gpu_row_refs_p2p.py)
                feeder.quiesce()
                feeder.devit = None
                gc.collect()
                dist.barrier()
                refs_sampler = FastSampler(4, a.slots, cfg, row_refs=True)

                def make_refs_iter(idx):
                    refs_sampler.idx = idx
                    return DeviceDistributedPrefetcher([dev], iter(refs_sampler), pipeline_on=True)
                refs_feeder = EpochFeeder(make_refs_iter, shuffler, get_idx)
                r_step = make_model_step(F, n_classes, a.hidden, layers, hip=True, arch=a.model, ddp=True)
                r_only, r_data, r_detail = model_step_timing(refs_feeder, r_step)
                refs_feeder.quiesce()
                t = torch.tensor([r_only, r_data], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                r_only, r_data = (float(v) for v in t.cpu().tolist())
                model_out["model_step"]["fused_first_layer"] = {
                    "what": "Session(row_refs): the delivery writes one address per row (+ a contiguous copy of the rows received "
                            "over RCCL), SAGE layer 1 reads the rows where they live (spp_sage_operand_forward_rows)",
                    "ms_per_step_model_only_resident_batch": r_only, "ms_per_step_with_data_path": r_data,
                    "data_path_cost_ms": r_data - r_only, "epoch_time_s_with_model_step": nb_epoch * r_data / 1e3,
                    "timing": r_detail}
                model_out["model_step"]["data_path_cost_ms"] = m_data - m_only
                if epoch_measured is not None:
                    refs_feeder.devit = None
                    gc.collect()
                    epoch_measured["with_model_step_fused_first_layer"] = run_epochs(make_refs_iter, r_step)
                refs_feeder.devit = None
                del r_step
                gc.collect()
            if not distributed and a.model == "sage" and not a.no_fused_leg:
                # Row g1: the opt-in fused consumer.  The Session delivers MFG + labels + n_id and NO feature rows
                # (PreparedBatch.x = TableRows(resident table, n_id)); models.SAGE's first layer aggregates straight from
                # the table (bit-identical operand, tests/test_gpu_model_step.py).  Same model, same optimiser, same
                # windows as the default legs above, which stay the ones `epoch_time_s_with_model_step` is quoted on.
                feeder.devit = None                   # the default Session ends: its pooled sampler serves the next one
                gc.collect()
                fused_sampler = FastSampler(4, a.slots, cfg, table_features=True)

                def make_fused_iter(idx):
                    fused_sampler.idx = idx
                    return DevicePrefetcher([dev], iter(fused_sampler))
                fused_feeder = EpochFeeder(make_fused_iter, shuffler, shuffler.get_idx)
                fused_step = make_model_step(F, n_classes, a.hidden, layers, hip=True, arch=a.model)
                f_only, f_data, f_detail = model_step_timing(fused_feeder, fused_step)
                fused_feeder.devit = None
                gc.collect()
                model_out["model_step"]["fused_first_layer"] = {
                    "what": "Session(table_features): no x gather in the delivery, SAGE layer 1 reads table[n_id[j]] itself "
                            "(spp_sage_operand_forward_table)",
                    "ms_per_step_model_only_resident_batch": f_only, "ms_per_step_with_data_path": f_data,
                    "data_path_cost_ms": f_data - f_only, "epoch_time_s_with_model_step": nb_epoch * f_data / 1e3,
                    "timing": f_detail}
                model_out["model_step"]["data_path_cost_ms"] = m_data - m_only
                if epoch_measured is not None:
                    _trace("measuring whole epochs (fused consumer)")
                    epoch_measured["with_model_step_fused_first_layer"] = run_epochs(make_fused_iter, fused_step)
                    gc.collect()
                del fused_step
            if not distributed:
                t_step = make_model_step(F, n_classes, a.hidden, layers, hip=False)
                t_only, t_data, _ = model_step_timing(feeder, t_step, windows=2, warm=4)
                model_out["model_step"]["plain_torch_formulation"] = {
                    "ms_per_step_model_only_resident_batch": t_only, "ms_per_step_with_data_path": t_data}
        except Exception as e:  # noqa: BLE001
            if distributed:
                raise                                 # a rank that fell out of a collective sequence: fail loudly
            model_out = {"model_step": {"error": repr(e)}}
    # ---- opt-in second leg: the same partitioned workload over the P2P transport ----
    p2p_out = None
    if distributed and a.p2p_leg:
        feeder.quiesce()
        feeder.devit = None
        gc.collect()
        dist.barrier()
        transport_before = os.environ.get("SPP_DIST_TRANSPORT")
        os.environ["SPP_DIST_TRANSPORT"] = "p2p"
        try:
            import dataclasses
            p2p_sampler = FastSampler(4, a.slots, cfg)

            def make_p2p_iter(idx):
                p2p_sampler.idx = idx
                return DeviceDistributedPrefetcher([dev], iter(p2p_sampler), pipeline_on=True)
            # parity first: a group of batches against the full table every rank holds
            n_check = min(8, max(1, n_local // bs))
            vcfg = dataclasses.replace(cfg, idx=get_idx()[:n_check * bs].contiguous(), exact_num_batches=n_check)
            vit = iter(FastSampler(4, a.slots, vcfg))            # (collective: the peers' partitions are mapped here)
            good = bool(vit.session.p2p)
            for proto in vit:
                good = good and proto.x is not None and bool(torch.equal(proto.x, wl.x[proto.n_id]))
            vit.session.close()
            del vit
            flags = torch.tensor([1 if good else 0], device=dev)
            dist.all_reduce(flags, op=dist.ReduceOp.MIN)
            p_feeder = EpochFeeder(make_p2p_iter, shuffler, get_idx)
            for _ in range(a.prime + a.warmup):
                p_feeder.next()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            pw = []
            for _w in range(max(2, R // 2)):
                e_ = 0
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    e_ += count_edges(p_feeder.next())
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                pw.append((time.perf_counter() - t0, float(e_)))
            st_ = torch.tensor(pw, dtype=torch.float64, device=dev)
            tmax = st_[:, 0].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            etot = st_[:, 1].clone()
            dist.all_reduce(etot, op=dist.ReduceOp.SUM)
            p_dt = float(tmax.mean())
            p2p_out = {"transport": "P2P: remote rows read in the owner's partition inside the delivery (no id exchange, no serve gather, "
                                    "no send / receive buffers); peers mapped through hipIpcOpenMemHandle",
                       "verified_bit_exact_vs_full_table": bool(int(flags.item())),
                       "windows": len(pw), "steps_each": a.steps, "ms_per_step": p_dt / a.steps * 1e3,
                       "value": float(etot.mean()) / p_dt, "ms_per_step_all": [round(float(v) / a.steps * 1e3, 5) for v in tmax.cpu().tolist()],
                       "vs_rccl": (dt / a.steps) / (p_dt / a.steps)}
            p_feeder.devit = None
            gc.collect()
            dist.barrier()
            if not a.no_model_step and a.model == "sage" and not a.no_fused_leg:
                # ... and the consumer this transport is meant for: row references on top (remote rows are not even copied:
                # the first layer reads them in their owners' HBM), under DistributedDataParallel like the default legs
                os.environ["SPP_EXCHANGE_ISSUE"] = "consumer"
                pr_sampler = FastSampler(4, a.slots, cfg, row_refs=True)

                def make_pr_iter(idx):
                    pr_sampler.idx = idx
                    return DeviceDistributedPrefetcher([dev], iter(pr_sampler), pipeline_on=True)
                pr_feeder = EpochFeeder(make_pr_iter, shuffler, get_idx)
                pr_step = make_model_step(F, 47, a.hidden, a.layers if a.layers > 0 else len(sizes), hip=True, arch=a.model, ddp=True)
                pr_only, pr_data, pr_detail = model_step_timing(pr_feeder, pr_step)
                t = torch.tensor([pr_only, pr_data], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                pr_only, pr_data = (float(v) for v in t.cpu().tolist())
                p2p_out["model_step_row_refs"] = {"ms_per_step_model_only_resident_batch": pr_only, "ms_per_step_with_data_path": pr_data,
                                                  "data_path_cost_ms": pr_data - pr_only, "timing": pr_detail}
                pr_feeder.devit = None
                del pr_step
                gc.collect()
                dist.barrier()
        finally:
            if transport_before is None:
                os.environ.pop("SPP_DIST_TRANSPORT", None)
            else:
                os.environ["SPP_DIST_TRANSPORT"] = transport_before
    if epoch_measured is not None:
        # measured against extrapolated (batches x ms/step of the windows), per leg
        ext = {"data_path_only": nb_epoch * (dt / a.steps)}
        if model_out is not None and "epoch_time_s_with_model_step" in model_out:
            ext["with_model_step"] = model_out["epoch_time_s_with_model_step"]
            fl = model_out["model_step"].get("fused_first_layer")
            if fl:
                ext["with_model_step_fused_first_layer"] = fl["epoch_time_s_with_model_step"]
        for k_, v_ in epoch_measured.items():
            if k_ in ext and v_.get("steady_s_mean"):
                v_["extrapolated_s"] = ext[k_]
                v_["measured_over_extrapolated"] = v_["steady_s_mean"] / ext[k_]

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0 and not legs_state["partial"]:
        emit_line()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
