"""ctypes binding of libspp_hip.so (the C ABI declared in include/spp.h).

There is no CPU fallback: if the library is missing or no HIP device is usable, the
product path raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspp_hip.so")

SPP_MAX_HOPS = 8
SPP_MAX_PARTS = 64

p = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64
u32 = C.c_uint32


class PartitionCfg(C.Structure):
    _fields_ = [("num_parts", i32), ("rank", i32), ("offsets", i64 * (SPP_MAX_PARTS + 1)),
                ("use_cache", i32), ("cache_map_dev", p), ("cache_map_len", i64)]


class SamplerOpts(C.Structure):
    """spp_sampler_opts: 0 = automatic, > 0 on, < 0 off (include/spp.h)."""
    _fields_ = [("col32", i32), ("deg_tags", i32), ("row_stubs", i32), ("rng_arena", i32), ("rng_arena_mb", i64),
                ("fuse_scatter", i32), ("flag_tiled", i32), ("rows_coalesced", i32), ("dedup_preread", i32),
                ("fuse_max_edges", i64), ("initial_edge_cap", i64), ("reserved", i64 * 4)]


class SamplerCfg(C.Structure):
    _fields_ = [("rowptr_dev", p), ("col_dev", p), ("num_nodes", i64), ("nnz", i64),
                ("num_hops", i32), ("sizes", i64 * SPP_MAX_HOPS), ("max_batch", i64),
                ("num_slots", i32), ("device", i32), ("replace", i32), ("part", PartitionCfg),
                ("graph_generation", i64), ("opts", SamplerOpts)]


class SamplerInfo(C.Structure):
    _fields_ = [("col32", i32), ("deg_tags", i32), ("row_stubs", i32), ("rng_arena", i32), ("idbits", i32),
                ("tag_cap", i32), ("num_hops", i32), ("dedup_buckets_log2", i32), ("dedup_table_slots", i32),
                ("generic", i32 * SPP_MAX_HOPS), ("fused_pick", i32 * SPP_MAX_HOPS), ("flag_tiled", i32 * SPP_MAX_HOPS),
                ("rows_coalesced", i32 * SPP_MAX_HOPS), ("bucket_log2", i32 * SPP_MAX_HOPS),
                ("col32_ms", C.c_double), ("row_stubs_ms", C.c_double), ("rng_arena_ms", C.c_double),
                ("col32_bytes", i64), ("row_stubs_bytes", i64), ("rng_arena_bytes", i64), ("rng_arena_batches", i64)]


class MfgCounts(C.Structure):
    _fields_ = [("num_nodes", i64), ("num_seeds", i64), ("num_hops", i32),
                ("T", i64 * SPP_MAX_HOPS), ("S", i64 * SPP_MAX_HOPS), ("E", i64 * SPP_MAX_HOPS),
                ("draws", i64), ("part_counts", i64 * (SPP_MAX_PARTS + 2))]


class MfgOut(C.Structure):
    _fields_ = [("n_id", p), ("rowptr", p * SPP_MAX_HOPS), ("col", p * SPP_MAX_HOPS),
                ("parts", p), ("cached", p), ("perm", p), ("row_addr", p), ("x_remote", p)]


class GroupOut(C.Structure):
    _fields_ = [("mfg", MfgOut), ("x_out", p), ("y_out", p)]


class ExchangeCfg(C.Structure):
    _fields_ = [("comm", p), ("x_local_dev", p), ("x_local_rows", i64), ("row_bytes", i64),
                ("cache_feats_dev", p), ("cache_rows", i64), ("x_local_stride_bytes", i64),
                ("cache_stride_bytes", i64), ("peer_x_dev", C.POINTER(p)), ("peer_x_stride_bytes", i64),
                ("issue_on_consumer", i32)]


class SessionCfg(C.Structure):
    _fields_ = [("rowptr_dev", p), ("col_dev", p), ("num_nodes", i64), ("nnz", i64),
                ("idx_dev", p), ("n_idx", i64), ("batch_size", i64), ("num_hops", i32),
                ("sizes", i64 * SPP_MAX_HOPS), ("skip_nonfull_batch", i32),
                ("force_exact_num_batches", i32), ("exact_num_batches", i64),
                ("max_items_in_queue", i32), ("group_size", i32), ("device", i32), ("sampler", p),
                ("part", C.POINTER(PartitionCfg)), ("exchange", C.POINTER(ExchangeCfg)),
                ("input_stream", p), ("order_after_input_stream", i32)]


class BatchDesc(C.Structure):
    _fields_ = [("batch_index", i64), ("start", i32), ("stop", i32), ("slot", i32),
                ("counts", MfgCounts)]


# name -> (restype, argtypes); every symbol include/spp.h declares
SIGNATURES = {
    "spp_abi_version": (C.c_int, []),
    "spp_last_error": (C.c_char_p, []),
    "spp_device_count": (C.c_int, []),
    "spp_async_errors": (C.c_int, [C.c_int, C.c_int]),
    "spp_profile_enable": (None, [C.c_int]),
    "spp_tune": (C.c_int, [C.c_char_p, C.c_int]),
    "spp_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(i64)]),
    "spp_mt19937_fill": (C.c_int, [u32, i64, i64, p, p]),
    "spp_batch_seed": (u32, [i32]),
    "spp_gather_rows": (C.c_int, [p, i64, i64, p, C.c_int, i64, i64, p, p]),
    "spp_gather_rows_strided": (C.c_int, [p, i64, i64, i64, p, C.c_int, i64, i64, p, p]),
    "spp_to_row_major": (C.c_int, [p, i64, i64, C.c_int, p, p]),
    "spp_sampler_create": (C.c_int, [C.POINTER(SamplerCfg), C.POINTER(p)]),
    "spp_sampler_destroy": (None, [p]),
    "spp_sampler_workspace_bytes": (i64, [p]),
    "spp_sampler_deliver_stream": (p, [p]),
    "spp_sampler_get_cfg": (C.c_int, [p, C.POINTER(SamplerCfg)]),
    "spp_sampler_get_info": (C.c_int, [p, C.POINTER(SamplerInfo)]),
    "spp_sampler_sample": (C.c_int, [p, i32, p, i64, u32, i64, p]),
    "spp_sampler_wait": (C.c_int, [p, i32, C.POINTER(MfgCounts)]),
    "spp_sampler_export": (C.c_int, [p, i32, C.POINTER(MfgOut), p]),
    "spp_sampler_gather": (C.c_int, [p, i32, p, i64, i64, i64, i64, p, p]),
    "spp_nid2partid": (C.c_int, [p, i32, p, i64, p, p]),
    "spp_cache_build_map": (C.c_int, [p, i64, p, i64, p]),
    "spp_cache_lookup": (C.c_int, [p, i64, p, i64, p, p, p]),
    "spp_partition_workspace_bytes": (i64, [i64]),
    "spp_partition_batch": (C.c_int, [p, i64, p, i32, i32, i32, p, i64, i64, p, p, p, p, p, p, i64, p]),
    "spp_assemble_features": (C.c_int, [p, p, i64, p, i32, i32, i64, p, i64, p, p, p, i64, i64, i64, p, p, p]),
    "spp_session_create": (C.c_int, [C.POINTER(SessionCfg), C.POINTER(p)]),
    "spp_session_destroy": (None, [p]),
    "spp_session_num_total_batches": (i64, [p]),
    "spp_session_num_consumed_batches": (i64, [p]),
    "spp_session_batch_ranges": (C.c_int, [p, p]),
    "spp_session_next": (C.c_int, [p, C.POINTER(BatchDesc)]),
    "spp_session_export": (C.c_int, [p, C.POINTER(MfgOut), p, i64, i64, i64, p, p, i64, i64, p, p]),
    "spp_session_next_group": (C.c_int, [p, i32, C.POINTER(BatchDesc), C.POINTER(i32)]),
    "spp_session_export_group": (C.c_int, [p, i32, C.POINTER(GroupOut), p, i64, i64, i64, p, i64, i64, p]),
    "spp_session_blocked_us": (i64, [p]),
    "spp_session_blocked_occasions": (i64, [p]),
    "spp_session_sampler": (p, [p]),
    "spp_session_group_size": (i32, [p]),
    "spp_comm_unique_id": (C.c_int, [p]),
    "spp_comm_create": (C.c_int, [p, i32, i32, i32, C.POINTER(p)]),
    "spp_comm_create_local": (C.c_int, [i32, i32, C.POINTER(p)]),
    "spp_comm_destroy": (None, [p]),
    "spp_comm_rank": (i32, [p]),
    "spp_comm_world": (i32, [p]),
    "spp_vip_frequencies": (C.c_int, [p, p, i64, p, i64, i64, p, i32, p, p, p]),
    "spp_csr_mean_forward": (C.c_int, [p, p, i64, p, i32, i64, i64, p, i64, p]),
    "spp_csr_mean_backward": (C.c_int, [p, p, i64, p, i64, i64, p, p]),
    "spp_sage_operand_forward": (C.c_int, [p, p, i64, p, i32, i64, i64, p, i64, p]),
    "spp_sage_operand_forward_table": (C.c_int, [p, p, i64, p, i32, i64, i64, p, i64, p, i64, p]),
    "spp_sage_operand_backward": (C.c_int, [p, p, i64, i64, p, i64, i64, p, p]),
    "spp_sage_operand_backward_workspace_bytes": (i64, [i64, i64, i64]),
    "spp_sage_operand_backward_gather": (C.c_int, [p, p, i64, i64, i64, p, i64, i64, p, p, i64, p]),
    "spp_relu_dropout_forward": (C.c_int, [p, i64, C.c_float, i32, C.c_uint64, p, p]),
    "spp_relu_dropout_backward": (C.c_int, [p, p, i64, C.c_float, p, p]),
    "spp_sage_operand_forward_act": (C.c_int, [p, p, i64, p, i64, p, i64, C.c_float, i32, C.c_uint64, p]),
    "spp_relu_dropout_backward_pre": (C.c_int, [p, p, i64, C.c_float, i32, C.c_uint64, p, p]),
    "spp_sage_operand_backward_gather_act": (C.c_int, [p, p, i64, i64, i64, p, i64, i64, p, p, i64, p, C.c_float, i32,
                                                       C.c_uint64, p]),
    "spp_gat_forward": (C.c_int, [p, p, i64, p, i64, p, p, C.c_float, p, p, p, p]),
    "spp_gat_logits": (C.c_int, [p, i32, i64, i64, i64, i64, p, p, p, p, p]),
    "spp_gat_logits_backward": (C.c_int, [p, i32, i64, i64, i64, i64, p, p, p, p, p]),
    "spp_gat_aggregate_forward": (C.c_int, [p, p, i64, p, i32, i64, i64, p, p, C.c_float, p, p, p, p]),
    "spp_gat_aggregate_backward": (C.c_int, [p, p, i64, p, i32, i64, i64, p, p, C.c_float, p, p, p, p, p, p, p, p]),
    "spp_gat_backward": (C.c_int, [p, p, i64, p, i64, p, p, C.c_float, p, p, p, p, p, p, p, p]),
    "spp_gat_aggregate_backward_gather_workspace_bytes": (i64, [i64, i64, i64]),
    "spp_gat_aggregate_backward_gather": (C.c_int, [p, p, i64, i64, i64, p, i32, i64, i64, p, p, C.c_float, p, p, p, p,
                                                    p, p, p, p, p, p, i64, p]),
    "spp_sage_operand_forward_rows": (C.c_int, [p, p, i64, p, i32, i64, p, i64, p]),
    "spp_gather_row_refs": (C.c_int, [p, i64, i64, p, p]),
    "spp_ipc_export": (C.c_int, [p, p, C.POINTER(i64)]),
    "spp_ipc_open": (C.c_int, [p, i32, C.POINTER(p)]),
    "spp_ipc_close": (C.c_int, [p]),
    "spp_session_try_next": (C.c_int, [p, C.POINTER(BatchDesc)]),
    "spp_session_quiesce": (C.c_int, [p]),
    "spp_session_exchange_stats": (C.c_int, [p, C.POINTER(i64), C.POINTER(i64)]),
}
SPP_COMM_ID_BYTES = 128
SPP_IPC_HANDLE_BYTES = 64

_lib = None


class SppError(RuntimeError):
    pass


def load():
    """Load libspp_hip.so and type every entry point.  Raises if the extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SppError(
                f"{LIB_PATH} is missing: build it with `python -m salient_plusplus_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as the
        # system one).  It must be resident BEFORE this library is loaded so both resolve to the same
        # runtime; loading /opt/rocm's copy first leaves torch without a usable device.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.spp_abi_version() != 6:
            raise SppError("libspp_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int):
    if rc < 0:
        msg = load().spp_last_error()
        raise SppError(msg.decode() if msg else f"libspp_hip error {rc}")
    return rc


def require_device():
    n = load().spp_device_count()
    if n <= 0:
        msg = load().spp_last_error()
        raise SppError("no usable HIP device for the MI355X data path: "
                       + (msg.decode() if msg else "device count 0"))
    return n
