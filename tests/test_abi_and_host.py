"""CPU-only: the C-ABI library loads and exports every symbol include/spp.h declares; host-side
logic of the facade (no compute calls, no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "spp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(spp_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from salient_plusplus_amd import _native as nat
    from salient_plusplus_amd import build
    build.build()
    L = nat.load()
    syms = declared_symbols()
    assert len(syms) >= 30
    for name in syms:
        assert hasattr(L, name), f"{name} declared in include/spp.h but not exported"
        assert name in nat.SIGNATURES, f"{name} has no ctypes signature"
    assert set(nat.SIGNATURES) == set(syms)
    assert L.spp_abi_version() == 6
    assert L.spp_batch_seed(64) == 64 * 17 + 5


def test_struct_layouts_match_header():
    """sizeof of the by-pointer structs, cross-checked by compiling the header with gcc."""
    import subprocess
    import tempfile
    from salient_plusplus_amd import _native as nat
    prog = r'''
#include <stdio.h>
#include "spp.h"
int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(spp_sampler_cfg), sizeof(spp_mfg_counts),
                        sizeof(spp_mfg_out), sizeof(spp_session_cfg), sizeof(spp_batch_desc), sizeof(spp_sampler_opts),
                        sizeof(spp_sampler_info), sizeof(spp_exchange_cfg), sizeof(spp_group_out)); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(nat.SamplerCfg), ctypes.sizeof(nat.MfgCounts), ctypes.sizeof(nat.MfgOut),
                     ctypes.sizeof(nat.SessionCfg), ctypes.sizeof(nat.BatchDesc), ctypes.sizeof(nat.SamplerOpts),
                     ctypes.sizeof(nat.SamplerInfo), ctypes.sizeof(nat.ExchangeCfg), ctypes.sizeof(nat.GroupOut)]


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from salient_plusplus_amd import _native as nat
    from salient_plusplus_amd import fast_sampler as fs
    with pytest.raises(nat.SppError):
        fs.serial_index(torch.zeros(4, 2), torch.zeros(1, dtype=torch.int64))
    cfg = fs.Config()
    cfg.rowptr, cfg.col, cfg.idx = torch.zeros(2, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), \
        torch.zeros(1, dtype=torch.int64)
    cfg.batch_size, cfg.sizes = 1, [1]
    with pytest.raises(nat.SppError):
        fs.Session(1, 1, cfg)
    # the rows either side of the path have no CPU fallback either
    from salient_plusplus_amd.fast_trainer.vip_cache import vip_frequencies
    from salient_plusplus_amd.models import mean_aggregate
    with pytest.raises(nat.SppError):
        vip_frequencies(cfg.rowptr, cfg.col, cfg.idx, [1], 1)
    with pytest.raises(nat.SppError):
        fs.NativeComm.local(2)
    with pytest.raises((nat.SppError, AssertionError)):
        mean_aggregate(torch.zeros(2, 4), torch.zeros(2, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), 1)


def test_local_transport_is_opt_in(monkeypatch):
    """The in-process rehearsal transport ships inside the product library for the test suite only: without
    SPP_ALLOW_LOCAL_COMM=1 (tests/conftest.py sets it) spp_comm_create_local refuses, with or without a GPU."""
    import ctypes as C
    from salient_plusplus_amd import _native as nat
    monkeypatch.delenv("SPP_ALLOW_LOCAL_COMM", raising=False)
    L = nat.load()
    arr = (C.c_void_p * 2)()
    assert L.spp_comm_create_local(2, 0, arr) == -1                      # SPP_ERR_INVALID, before any device is touched
    assert b"SPP_ALLOW_LOCAL_COMM" in L.spp_last_error() and not arr[0]


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under salient_plusplus_amd/ may import, link or
    execute it (no CPU fallback on the product path)."""
    pkg = os.path.join(ROOT, "salient_plusplus_amd")
    bad = re.compile(r"(^|\n)\s*(from|import)\s+oracle\b|liborc|spp_oracle|orc_[a-z_]+\(")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not bad.search(text), f"{os.path.join(dirpath, f)} references the oracle"


def test_config_facade_num_batches_and_field_set(golden_dir):
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSamplerConfig
    from oracle import oracle as orc
    base = dict(x_cpu=torch.zeros(1, 1), x_gpu=torch.empty(0), y=torch.zeros(1), rowptr=torch.zeros(2),
                col=torch.zeros(0), batch_size=64, sizes=[15, 10, 5], pin_memory=False, distributed=False,
                partition_book=None, cache=fs.Cache(), count_remote_frequency=False, use_cache=False)
    for n in (200, 128, 10, 64):
        for skip in (False, True):
            cfg = FastSamplerConfig(idx=torch.zeros(n, dtype=torch.int64), skip_nonfull_batch=skip,
                                    force_exact_num_batches=False, exact_num_batches=0, **base)
            assert cfg.get_num_batches() == orc.batch_ranges(n, 64, skip).shape[0]
    cfg = FastSamplerConfig(idx=torch.zeros(300, dtype=torch.int64), skip_nonfull_batch=False,
                            force_exact_num_batches=True, exact_num_batches=4, **base)
    assert cfg.get_num_batches() == 4 == orc.batch_ranges(300, 64, False, True, 4).shape[0]
    native = cfg.to_fast_sampler()
    # the rw fields of the reference's Config (fast_sampler.cpp:1290-1309)
    for f in ["x_cpu", "x_gpu", "y", "rowptr", "col", "idx", "batch_size", "sizes", "skip_nonfull_batch",
              "pin_memory", "distributed", "partition_book", "cache", "force_exact_num_batches",
              "exact_num_batches", "count_remote_frequency", "use_cache"]:
        assert hasattr(native, f)
    assert isinstance(native.partition_book, fs.RangePartitionBook)   # skipped when not distributed


def test_adj_mapping_and_batch_records():
    from salient_plusplus_amd.fast_trainer.samplers import (Adj__from_fast_sampler, PreparedBatch,
                                                             ProtoDistributedBatch)
    rowptr = torch.tensor([0, 2, 3]); col = torch.tensor([0, 2, 1]); e_id = torch.empty(0, dtype=torch.int64)
    adj = Adj__from_fast_sampler((rowptr, col, e_id, (2, 3)))
    assert tuple(adj.size) == (3, 2)                       # sparse_sizes[::-1]
    assert tuple(adj.adj_t.sparse_sizes()) == (2, 3)
    rp, cl, val = adj.adj_t.csr()
    assert torch.equal(rp, rowptr) and torch.equal(cl, col) and val is None
    pb = PreparedBatch.from_fast_sampler((torch.zeros(3, 4), torch.arange(2).unsqueeze(-1), [(rowptr, col, e_id, (2, 3))], (5, 7)))
    assert pb.batch_size == 2 and pb.num_total_nodes == 3 and pb.y.shape == (2,)
    assert pb.idx_range == slice(5, 7)
    pb.record_stream(None)                                 # CPU tensors: no-op
    pb2 = pb.to("cpu")
    assert torch.equal(pb2.x, pb.x)

    class Raw:
        partition_nids = [torch.tensor([1, 2]), torch.tensor([7])]
        sliced_cpu_features = torch.empty(0, 4)
        sliced_cpu_labels = torch.tensor([[1], [0]])
        cached_nids = torch.empty(0, dtype=torch.int64)
        perm_partition_to_mfg = torch.tensor([0, 2, 1])
        adjs = [(rowptr, col, e_id, (2, 3))]
        idx_range = (0, 2)
    proto = ProtoDistributedBatch.from_fast_sampler(Raw)
    assert proto.num_total_nodes == 3 and proto.num_cached_nodes == 0
    assert proto.get_num_local_nodes(0) == 2 and proto.get_num_communicated_nodes(0) == 1
    assert proto.idx_range == slice(0, 2)

    # the statistics form of the batch (reference samplers.py:167-196) and the timer hook (transferers.py:13-18)
    from salient_plusplus_amd.fast_trainer import transferers as tr
    from salient_plusplus_amd.fast_trainer.samplers import NumpyProtoDistributedBatch
    assert NumpyProtoDistributedBatch._fields == ("partition_nids", "cache_specific_nids", "perm_partition_to_mfg", "adjs", "seed_indices")
    npb = NumpyProtoDistributedBatch.from_proto_batch(proto, torch.tensor([40, 41, 42]))
    assert [a.tolist() for a in npb.partition_nids] == [[1, 2], [7]] and npb.seed_indices.tolist() == [40, 41]
    assert npb.adjs[0].shape == (2, 3) and npb.adjs[0].indptr.tolist() == [0, 2, 3] and npb.adjs[0].indices.tolist() == [0, 2, 1]

    class Tick:
        name = "stage"
        nanos = 5
    tr.aggregate_time_results.clear()
    tr.aggregate_time(Tick); tr.aggregate_time(Tick)
    assert tr.aggregate_time_results == {"stage": 10}
    tr.aggregate_time_results.clear()


def test_shufflers_follow_reference_seeding():
    from salient_plusplus_amd.fast_trainer.shufflers import DistributedShuffler, Shuffler
    idx = torch.arange(100, 200)
    s = Shuffler(idx)
    s.set_epoch(3)
    g = torch.Generator(device="cpu")
    g.manual_seed(2147483647 + 3)                          # shufflers.py:25-29
    want = idx[torch.randperm(100, generator=g)]
    assert torch.equal(s.get_idx(), want)
    d = DistributedShuffler(idx, 4)
    d.set_epoch(3)
    parts = [d.get_idx(r) for r in range(4)]
    assert torch.equal(torch.cat(parts), want)
    assert [p.numel() for p in parts] == [25, 25, 25, 25]


def test_range_partition_book_host_side(golden_dir):
    from salient_plusplus_amd import fast_sampler as fs
    p = np.load(os.path.join(golden_dir, "partition_book.npz"))
    pb = fs.RangePartitionBook(2, 4, torch.from_numpy(p["offsets"]))
    nids = torch.from_numpy(p["nids"])
    if torch.cuda.is_available():
        np.testing.assert_array_equal(pb.nid2partid(nids).numpy(), p["partid"])
    else:   # the ownership lookup has ONE implementation, on the GPU: without one it refuses
        from salient_plusplus_amd import _native as nat
        with pytest.raises(nat.SppError):
            pb.nid2partid(nids)
    np.testing.assert_array_equal(pb.nid2localnid(nids, 2).numpy(), p["localnid_p2"])
    np.testing.assert_array_equal(pb.partid2nids(1).numpy(), p["partid2nids_1"])
    assert pb.nid_is_local(torch.tensor([1499, 1500, 2099, 2100])).tolist() == [False, True, True, False]


def test_bench_refuses_to_run_fewer_ranks_than_asked_for():
    """`python bench.py --gpus N` starts its own ranks (children, before anything touches the GPU); on a box with
    fewer GPUs it says so and exits non-zero within seconds instead of measuring one GPU and calling it N."""
    import subprocess
    import sys
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "refusing to run fewer ranks" in r.stderr
    assert r.stdout.strip() == ""                       # no JSON line that could be mistaken for a measurement
    assert time.time() - t0 < 60
    # a launcher that started another number of ranks than --gpus says is refused as well
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env2, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_bench_launcher_never_loads_the_gpu_runtime(tmp_path):
    """The parent of `bench.py --gpus N`'s rank processes counts GPUs from the KFD topology in sysfs and is still free of
    torch / libamdhip64 when it would spawn (--launch-dry-run reports /proc/self/maps instead of starting anything)."""
    import json
    import subprocess
    import sys
    topo = tmp_path / "nodes"
    for k, props in enumerate(("cpu_cores_count 64\nsimd_count 0\n", "simd_count 1024\ndrm_render_minor 128\n",
                               "simd_count 1024\ndrm_render_minor 129\n", "simd_count 1024\ndrm_render_minor 130\n")):
        (topo / str(k)).mkdir(parents=True)
        (topo / str(k) / "properties").write_text(props)
    base = {k: v for k, v in os.environ.items()
            if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
    base["SPP_KFD_TOPOLOGY"] = str(topo)

    def dry(gpus, **env):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--launch-dry-run"],
                           env=dict(base, **env), capture_output=True, text=True, timeout=60)
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, (r.stdout, r.stderr[-500:])
        return r.returncode, json.loads(lines[0]), r.stderr

    rc, d, _ = dry(2)
    assert rc == 0 and d["visible_gpus"] == 3 and d["ranks_wanted"] == 2            # the CPU node is not counted
    assert d["torch_imported"] is False and d["libamdhip64_mapped"] is False and d["libhsa_runtime_mapped"] is False
    rc, d, err = dry(4)
    assert rc == 2 and "refusing to run fewer ranks" in err and d["libamdhip64_mapped"] is False
    # the runtime's own narrowing is honoured: ROCR first, HIP among what is left; an empty list hides every GPU
    assert dry(2, HIP_VISIBLE_DEVICES="0,2")[1]["visible_gpus"] == 2
    assert dry(2, ROCR_VISIBLE_DEVICES="1,2", HIP_VISIBLE_DEVICES="1")[1]["visible_gpus"] == 1
    assert dry(2, HIP_VISIBLE_DEVICES="0,7,1")[1]["visible_gpus"] == 1               # stops at the first index that does not exist
    rc, d, _ = dry(2, HIP_VISIBLE_DEVICES="")
    assert rc == 2 and d["visible_gpus"] == 0


def test_table_rows_reads_like_the_feature_matrix():
    """fast_sampler.TableRows (row g1: the batch's features as (resident table, n_id)): sizes, dtype and device read like
    the matrix x = table[n_id] the reference delivers (fast_sampler.cpp:1004-1016), and the façade's opt-in keyword
    defaults to the reference behaviour.  (materialize() and the fused first layer are -m gpu tests.)"""
    import dataclasses

    import torch
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    storage = torch.zeros((50, 16), dtype=torch.float16)
    table = storage[:, :12]                                   # a strided view, as the padded resident table is
    n_id = torch.tensor([3, 49, 3, 0, 7], dtype=torch.int64)
    t = fs.TableRows(table, n_id)
    assert t.shape == torch.Size((5, 12)) and t.size(0) == 5 and t.size(1) == 12 and tuple(t.size()) == (5, 12)
    assert t.dim() == 2 and t.numel() == 60 and t.dtype == torch.float16
    assert t.is_cuda is False and t.device == n_id.device
    assert t.to(None) is t and t.to(n_id.device) is t and t.to(device=torch.device("cpu"), non_blocking=True) is t
    t.record_stream(None)                                     # host tensors: nothing to record
    f = {fld.name: fld for fld in dataclasses.fields(FastSampler)}
    assert list(f)[:3] == ["num_threads", "max_items_in_queue", "cfg"]           # the reference's three (samplers.py:381-399)
    assert f["table_features"].default is False


def test_sampler_options_and_p2p_stride_rules_host_side():
    """Host-only logic of round 6: option names are checked, booleans map to +1 / -1, the context manager restores; the row
    stride of a one-row partition follows the resident tables' rule; peers with different strides are refused."""
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd import _native as nat
    fs.set_sampler_options()
    with pytest.raises(RuntimeError):
        fs.set_sampler_options(no_such_option=1)
    with fs.sampler_options(row_stubs=False, rng_arena=True, fuse_scatter=2, col32=0):
        assert fs._sampler_opts == {"row_stubs": -1, "rng_arena": 1, "fuse_scatter": 2}
        with fs.sampler_options(deg_tags=False):
            assert fs._sampler_opts == {"deg_tags": -1}
        assert fs._sampler_opts == {"row_stubs": -1, "rng_arena": 1, "fuse_scatter": 2}
    assert fs._sampler_opts == {}
    assert set(fs._OPT_FIELDS) | {"reserved"} == {n for n, _t in nat.SamplerOpts._fields_}
    one = torch.zeros((1, 200), dtype=torch.float16)              # 400-byte rows live 512 bytes apart in a resident table
    many = torch.zeros((5, 256), dtype=torch.float16)[:, :200]
    assert fs._table_stride_bytes(one) == 512 == fs._table_stride_bytes(many)
    assert fs._table_stride_bytes(torch.zeros((1, 128), dtype=torch.float16)) == 256
    assert fs._common_stride([512, 512]) == 512
    with pytest.raises(RuntimeError):
        fs._common_stride([512, 400])
    r = fs.RowRefs(torch.zeros(7, dtype=torch.int64), None, 12, torch.float16, None, ())
    assert tuple(r.shape) == (7, 12) and r.size(0) == 7 and r.dim() == 2 and r.numel() == 84 and r.dtype == torch.float16
    assert r.to() is r
