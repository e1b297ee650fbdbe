"""-m gpu: bench.py's line carries what the contract asks for, on both of its paths.

The single-GPU path and the partitioned path (RCCL communicator at world size 1, bucketing, native exchange, assembly, the
model-step leg under DistributedDataParallel with consumer-issued exchanges) run as the driver runs them -- a child process,
one JSON line on stdout -- on the small S-arxiv workload with a handful of steps (two windows of 24: three sampling
groups, so that the sampler's roofline entry has chains to time)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _run(extra, more_env=None):
    env = dict(os.environ)
    env.update(more_env or {})
    env.setdefault("MASTER_PORT", "29741")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "S-arxiv", "--steps", "24", "--warmup", "2",
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
    assert d["n_gpus"] == n_gpus and d["steps"] == 24 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # the line describes its own timed region: the reported ms_per_step IS the plain mean, ms_per_step x steps x windows
    # reproduces timed_region_s, every window is listed; the trimmed mean and the median are extra keys
    w = d["windows"]
    assert abs(d["ms_per_step"] * 1e-3 * d["steps"] * w["n"] - d["timed_region_s"]) <= 1e-6 + 1e-3 * d["timed_region_s"]
    assert abs(w["ms_per_step_mean"] - d["ms_per_step"]) <= 1e-9
    allw = sorted(w["ms_per_step_all"])
    assert len(allw) == w["n"]
    assert abs(d["ms_per_step"] - sum(allw) / len(allw)) <= 1e-4 * d["ms_per_step"] + 1e-5
    assert "ms_per_step_trimmed_mean" in w and "ms_per_step_median" in w
    assert len(w["deliver_us_all"]) == w["n"] and all(v >= 0 for v in w["deliver_us_all"])   # per-window in-situ delivery time
    # which windows carry two chains' work: 24 steps = 1.5 groups of 16, so every window opens 1 or 2 groups
    assert len(w["groups_opened_all"]) == w["n"] and sum(w["groups_opened_all"]) >= 1
    assert w["opens_two_groups"] == [k for k, n_ in enumerate(w["groups_opened_all"]) if n_ >= 2]
    assert abs(d["value"] - d["sampled_edges_per_batch"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    # the sampler's own roofline entry: algorithmic bytes of the run's batches over the chains' in-situ spans
    rs = d["roofline_sampler"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_batch", "chain_span_ms_per_batch",
              "chains_timed"):
        assert k in rs, k
    assert rs["bound"] == "hbm" and rs["unit"] == "GB/s" and 0 < rs["frac"] < 1 and rs["chains_timed"] >= 1
    assert abs(rs["frac"] - rs["achieved"] / rs["peak"]) < 1e-9
    # 24 T + 16 E + 8 dU per hop is at least 16 bytes per sampled edge
    assert rs["algorithmic_bytes_per_batch"] >= 16 * d["sampled_edges_per_batch"]


def _check_epochs(d, legs):
    """whole epochs timed wall-clock (not batches x ms/step), the first one apart; beside them the extrapolation they replace"""
    em = d["epoch_measured"]
    for leg in legs:
        e = em[leg]
        assert e["epochs"] == 4 and e["batches"] >= 1 and e["first_s"] > 0 and len(e["steady_s_all"]) == 3
        assert all(v > 0 for v in e["steady_s_all"])
        assert abs(e["steady_s_mean"] - sum(e["steady_s_all"]) / 3) < 1e-4
        assert e["extrapolated_s"] > 0 and abs(e["measured_over_extrapolated"] - e["steady_s_mean"] / e["extrapolated_s"]) < 1e-3
    # the once-per-process work the windows never pay, and which chain variant the library chose by itself
    st = d["setup"]
    for k in ("col32_ms", "row_stubs_ms", "rng_arena_ms", "col32_GB", "row_stubs_GB", "rng_arena_GB", "rng_arena_batches"):
        assert k in st, k
    assert st["col32_GB"] > 0 and st["rng_arena_GB"] > 0 and st["rng_arena_ms"] > 0
    v = d["config"]["sampler_variant"]
    assert v["col32"] == 1 and v["row_stubs"] == 1 and v["deg_tags"] == 1 and v["rng_arena"] == 1
    assert v["fused_pick"] == [1, 1, 0] and v["flag_tiled"] == [0, 0, 1] and v["rows_coalesced"] == [1, 1, 1]


def test_single_gpu_line():
    d = _run([])
    _check_common(d, 1)
    _check_epochs(d, ["data_path_only", "with_model_step", "with_model_step_fused_first_layer"])
    assert d["model_step"]["hidden"] == 256 and d["model_step"]["layers"] == 3
    cpu = d["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    m = d["model_step"]
    assert m["ms_per_step_with_data_path"] > 0 and d["epoch_time_s_with_model_step"] > 0
    # row g1: both consumers in the line under distinct keys -- the default (x delivered) and the fused first layer
    f = m["fused_first_layer"]
    assert f["ms_per_step_with_data_path"] > 0 and f["ms_per_step_model_only_resident_batch"] > 0
    assert abs(f["data_path_cost_ms"] - (f["ms_per_step_with_data_path"] - f["ms_per_step_model_only_resident_batch"])) < 1e-9
    assert abs(m["data_path_cost_ms"] - (m["ms_per_step_with_data_path"] - m["ms_per_step_model_only_resident_batch"])) < 1e-9


def test_partitioned_path_line_with_the_ddp_leg():
    d = _run(["--gpus", "1", "--force-distributed", "--no-cpu-baseline", "--p2p-leg"])
    _check_common(d, 1)
    assert "range-partitioned" in d["config"]["parallelism"]
    m = d["model_step"]
    assert "DistributedDataParallel" in m["model"] and m["ms_per_step_with_data_path"] > 0
    ex = d.get("exchange")
    assert ex is not None and ex.get("rccl_world") == 1
    _check_epochs(d, ["data_path_only", "with_model_step", "with_model_step_fused_first_layer"])
    # row g1 where the metric lives: the partitioned path's fused consumer (row references) under its own key
    f = m["fused_first_layer"]
    assert "row_refs" in f["what"] and f["ms_per_step_with_data_path"] > 0
    assert abs(f["data_path_cost_ms"] - (f["ms_per_step_with_data_path"] - f["ms_per_step_model_only_resident_batch"])) < 1e-9
    # the opt-in P2P leg: parity checked against the full table, figures beside (not instead of) the RCCL line's
    p2p = d["exchange_p2p"]
    assert p2p["verified_bit_exact_vs_full_table"] is True and p2p["ms_per_step"] > 0 and p2p["value"] > 0
    assert p2p["model_step_row_refs"]["ms_per_step_with_data_path"] > 0


def test_model_shape_flags():
    """--hidden / --layers: BASELINE configs[4] names a 1024-wide SAGE over a two-hop fan-out"""
    d = _run(["--hidden", "1024", "--fanouts", "25,15", "--no-cpu-baseline", "--no-fused-leg", "--epochs", "0"])
    m = d["model_step"]
    assert m["hidden"] == 1024 and m["layers"] == 2 and "SAGE 2x1024" in m["model"] and m["ms_per_step_with_data_path"] > 0
    assert d["epoch_measured"] is None


def test_partitioned_line_survives_an_optional_leg_that_never_returns():
    """N > 1 (here: the partitioned path at world size 1): the legs behind the timed windows are sequences of collectives; if one
    hangs, a watchdog prints the line with what the windows measured and ends the process (SPP_BENCH_LEG_TIMEOUT_S)."""
    d = _run(["--gpus", "1", "--force-distributed", "--no-cpu-baseline"], {"SPP_BENCH_LEG_TIMEOUT_S": "0.2"})
    assert d["optional_legs_timed_out"] is True and d["value"] > 0 and d["ms_per_step"] > 0 and d["roofline"]["frac"] > 0
    assert "range-partitioned" in d["config"]["parallelism"]
