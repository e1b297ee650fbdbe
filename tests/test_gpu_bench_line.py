"""-m gpu: bench.py's line carries what the contract asks for, on both of its paths.

The single-GPU path and the partitioned path (RCCL communicator at world size 1, bucketing, native exchange, assembly, the
model-step leg under DistributedDataParallel with consumer-issued exchanges) run as the driver runs them -- a child process,
one JSON line on stdout -- on the small S-arxiv workload with a handful of steps."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _run(extra):
    env = dict(os.environ)
    env.setdefault("MASTER_PORT", "29741")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "S-arxiv", "--steps", "6", "--warmup", "2",
                        "--windows", "2", "--prime", "8", "--cpu-seconds", "1"] + extra, env=env, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line
    return json.loads(lines[0])


def _check_common(d, n_gpus):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "timed_region_s"):
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # the line describes its own timed region: the plain mean x steps x windows reproduces it, every window is listed,
    # and the reported figure is that mean (fewer than 8 windows) or the mean without the slowest and the fastest window
    w = d["windows"]
    assert abs(w["ms_per_step_mean"] * 1e-3 * d["steps"] * w["n"] - d["timed_region_s"]) <= 1e-6 + 1e-3 * d["timed_region_s"]
    allw = sorted(w["ms_per_step_all"])
    assert len(allw) == w["n"]
    kept = allw[1:-1] if w["n"] >= 8 else allw
    assert abs(d["ms_per_step"] - sum(kept) / len(kept)) <= 1e-4 * d["ms_per_step"] + 1e-5
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9


def test_single_gpu_line():
    d = _run([])
    _check_common(d, 1)
    cpu = d["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    m = d["model_step"]
    assert m["ms_per_step_with_data_path"] > 0 and d["epoch_time_s_with_model_step"] > 0


def test_partitioned_path_line_with_the_ddp_leg():
    d = _run(["--gpus", "1", "--force-distributed", "--no-cpu-baseline"])
    _check_common(d, 1)
    assert "range-partitioned" in d["config"]["parallelism"]
    m = d["model_step"]
    assert "DistributedDataParallel" in m["model"] and m["ms_per_step_with_data_path"] > 0
    ex = d.get("exchange")
    assert ex is not None and ex.get("rccl_world") == 1
