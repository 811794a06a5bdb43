"""ctypes binding of libleann_mi355x.so (the C ABI declared in include/leann_mi355x.h).

The library is built in-tree by ``leann_amd.build.build_all()`` (hipcc, gfx950).  There is no CPU
fallback: if the shared object is missing or no HIP device is visible, every compute entry point
raises -- loudly.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libleann_mi355x.so"
BUILD_LIB_PATH = _PKG / "lib" / "libleann_hnswbuild.so"

LM_OK, LM_EINVAL, LM_ENOENT, LM_EFORMAT, LM_EHIP, LM_ESTATE, LM_EPROVIDER = 0, -1, -2, -3, -4, -5, -6
METRIC_INNER_PRODUCT, METRIC_L2 = 0, 1
DTYPE_F32, DTYPE_F16 = 0, 1


class LeannMi355xError(RuntimeError):
    pass


class IndexInfo(C.Structure):
    _fields_ = [
        ("ntotal", C.c_int64), ("d", C.c_int32), ("d_padded", C.c_int32), ("metric", C.c_int32),
        ("entry_point", C.c_int32), ("max_level", C.c_int32), ("max_degree0", C.c_int32),
        ("max_degree_up", C.c_int32), ("n_neighbors", C.c_int64), ("has_table", C.c_int32),
        ("has_provider", C.c_int32), ("device", C.c_int32),
    ]


class SearchParams(C.Structure):
    """Mirror of lm_search_params == faiss.SearchParametersHNSW as filled in hnsw_backend.py:203-234."""

    _fields_ = [
        ("efSearch", C.c_int32), ("beam_size", C.c_int32), ("check_relative_distance", C.c_int32),
        ("pq_pruning_ratio", C.c_float), ("local_prune", C.c_int32), ("send_neigh_times_ratio", C.c_float),
        ("batch_size", C.c_int32), ("zmq_port", C.c_int32), ("recompute", C.c_int32), ("max_batch", C.c_int32),
        ("recompute_memo", C.c_int32),
    ]


class SearchStats(C.Structure):
    _fields_ = [
        ("ndis", C.c_int64), ("nunique", C.c_int64), ("nrounds", C.c_int64), ("nexpand", C.c_int64),
        ("update_launches", C.c_int64), ("update_ms", C.c_double), ("expand_ms", C.c_double),
        ("provider_ms", C.c_double), ("nadc", C.c_int64), ("update_span_ms", C.c_double), ("update_span_launches", C.c_int64),
    ]


class PqSearchParams(C.Structure):
    """Mirror of lm_pq_search_params == the batch_search argument list of diskann_backend.py:453-467."""

    _fields_ = [
        ("complexity", C.c_int32), ("beam_width", C.c_int32), ("num_threads", C.c_int32),
        ("use_deferred_fetch", C.c_int32), ("skip_search_reorder", C.c_int32), ("recompute_neighbors", C.c_int32),
        ("dedup_node_dis", C.c_int32), ("prune_ratio", C.c_float), ("batch_recompute", C.c_int32),
        ("use_global_pruning", C.c_int32),
    ]


class BertH384Layer(C.Structure):  # include/leann_mi355x.h: lm_bert_h384_layer
    _fields_ = [(n, C.c_void_p) for n in ("wqkv", "bqkv", "wo_img", "bo", "ln1_gamma", "ln1_beta", "w1_img", "b1", "w2_img", "b2", "ln2_gamma", "ln2_beta",
                                          "wo", "w1", "w2", "wqkv_img")]


class BertH384(C.Structure):  # include/leann_mi355x.h: lm_bert_h384
    _fields_ = [("n_layers", C.c_int32), ("heads", C.c_int32), ("ffn", C.c_int32), ("normalize", C.c_int32), ("pooling", C.c_int32), ("ln_eps", C.c_float),
                ("word", C.c_void_p), ("pos_table", C.c_void_p), ("type0", C.c_void_p), ("emb_gamma", C.c_void_p), ("emb_beta", C.c_void_p),
                ("layers", C.POINTER(BertH384Layer))]


class BertLayer(C.Structure):  # include/leann_mi355x.h: lm_bert_layer (plain nn.Linear weights)
    _fields_ = [(n, C.c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_gamma", "ln1_beta", "w1", "b1", "w2", "b2", "ln2_gamma", "ln2_beta")]


class Bert(C.Structure):  # include/leann_mi355x.h: lm_bert
    _fields_ = [("hidden", C.c_int32), ("n_layers", C.c_int32), ("heads", C.c_int32), ("ffn", C.c_int32), ("pooling", C.c_int32),
                ("normalize", C.c_int32), ("ln_eps", C.c_float),
                ("word", C.c_void_p), ("pos_table", C.c_void_p), ("type0", C.c_void_p), ("emb_gamma", C.c_void_p), ("emb_beta", C.c_void_p),
                ("layers", C.POINTER(BertLayer))]


class KernelTime(C.Structure):  # include/leann_mi355x.h: lm_kernel_time
    _fields_ = [("name", C.c_char_p), ("launches", C.c_int64), ("ms", C.c_double), ("work", C.c_double)]


KT_LAYER_TAIL, KT_GEMM_WS, KT_ATTN, KT_GEMM_F16, KT_QKV, KT_QKV_ATTN, KT_COUNT = 0, 1, 2, 3, 4, 5, 6


class RecomputeStats(C.Structure):  # include/leann_mi355x.h: lm_recompute_stats
    _fields_ = [(n, C.c_int64) for n in ("calls", "chunks", "tokens", "forwards", "host_syncs")]


ABI_REVISION = 6  # include/leann_mi355x.h: LM_ABI_REVISION

PROVIDER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_void_p)

# every symbol include/leann_mi355x.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "lm_last_error", "lm_device_count", "lm_version", "lm_abi_revision",
    "lm_index_read", "lm_index_create_from_csr", "lm_index_free", "lm_index_info",
    "lm_index_attach_table", "lm_index_set_provider", "lm_index_set_hub_cache", "lm_index_set_stream",
    "lm_search_params_default", "lm_index_search", "lm_index_search_device",
    "lm_index_get_stats", "lm_index_set_profiling", "lm_index_set_option", "lm_index_get_option", "lm_index_event_overhead_us",
    "lm_dist_gather", "lm_topk_merge",
    "lm_pq_attach", "lm_pq_attach_chunked", "lm_pq_search_params_default", "lm_pq_batch_search", "lm_pq_batch_search_device",
    "lm_add_layernorm_f16", "lm_attn_varlen_hd32_f16", "lm_attn_varlen_f16", "lm_embed_layernorm_f16", "lm_meanpool_varlen_f16",
    "lm_layer_tail_h384_f16", "lm_layer_tail_pack_h384", "lm_qkv_h384_f16", "lm_qkv_attn_h384_f16", "lm_h384_first_half_form", "lm_qkv_pack_h384", "lm_gemm_ws_h384_f16", "lm_gemm_f16", "lm_pack_tokens",
    "lm_tokens_create", "lm_tokens_free", "lm_tokens_gather", "lm_tokens_count",
    "lm_bert_h384_workspace_bytes", "lm_bert_h384_forward_packed",
    "lm_bert_workspace_bytes", "lm_bert_forward_packed", "lm_clspool_varlen_f16",
    "lm_recompute_create", "lm_recompute_create_general", "lm_recompute_free", "lm_recompute_provider", "lm_recompute_embed", "lm_recompute_get_stats", "lm_index_set_recompute",
    "lm_kernel_timing_enable", "lm_kernel_timing_read",
]

_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise LeannMi355xError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C leann_amd/csrc`. leann-backend-mi355x has no CPU fallback."
        )
    lib = C.CDLL(str(LIB_PATH))
    vp, i32, i64, f32p = C.c_void_p, C.c_int32, C.c_int64, C.c_void_p
    lib.lm_last_error.restype = C.c_char_p
    lib.lm_version.restype = C.c_char_p
    if lib.lm_abi_revision() != ABI_REVISION:  # a stale build of the library beside a newer package (or the reverse): struct layouts may differ
        raise LeannMi355xError(f"{LIB_PATH} speaks ABI revision {lib.lm_abi_revision()}, this package was written against {ABI_REVISION}: rebuild it "
                               "(`make -C leann_amd/csrc`)")
    lib.lm_device_count.restype = C.c_int
    lib.lm_index_read.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.lm_index_create_from_csr.argtypes = [i64, i32, i32, vp, vp, i64, vp, i64, vp, i32, i32, C.c_int, C.POINTER(vp)]
    lib.lm_index_free.argtypes = [vp]
    lib.lm_index_free.restype = None
    lib.lm_index_info.argtypes = [vp, C.POINTER(IndexInfo)]
    lib.lm_index_attach_table.argtypes = [vp, vp, i32, i64, i32, i32]
    lib.lm_index_set_provider.argtypes = [vp, PROVIDER_FN, vp]
    lib.lm_index_set_stream.argtypes = [vp, vp]
    lib.lm_index_set_hub_cache.argtypes = [vp, vp, i32, vp]
    lib.lm_search_params_default.argtypes = [C.POINTER(SearchParams)]
    lib.lm_search_params_default.restype = None
    lib.lm_index_search.argtypes = [vp, i64, f32p, i32, vp, vp, C.POINTER(SearchParams)]
    lib.lm_index_search_device.argtypes = [vp, i64, vp, i32, vp, vp, C.POINTER(SearchParams)]
    lib.lm_index_get_stats.argtypes = [vp, C.POINTER(SearchStats)]
    lib.lm_index_set_profiling.argtypes = [vp, i32]
    lib.lm_index_set_option.argtypes = [vp, C.c_char_p, i64]
    lib.lm_index_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    lib.lm_index_event_overhead_us.argtypes = [vp, C.POINTER(C.c_double)]
    lib.lm_dist_gather.argtypes = [vp, i32, i32, i32, vp, vp, vp, i64, vp, vp]
    lib.lm_topk_merge.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.lm_pq_attach.argtypes = [vp, i32, vp, vp, i64]
    lib.lm_pq_attach_chunked.argtypes = [vp, i32, vp, vp, vp, i64]
    lib.lm_pq_search_params_default.argtypes = [C.POINTER(PqSearchParams)]
    lib.lm_pq_search_params_default.restype = None
    lib.lm_pq_batch_search.argtypes = [vp, i64, vp, i32, C.POINTER(PqSearchParams), vp, vp]
    lib.lm_pq_batch_search_device.argtypes = [vp, i64, vp, i32, C.POINTER(PqSearchParams), vp, vp]
    lib.lm_add_layernorm_f16.argtypes = [vp, vp, vp, vp, vp, i64, i32, C.c_float, vp]
    lib.lm_attn_varlen_hd32_f16.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.lm_attn_varlen_f16.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.lm_embed_layernorm_f16.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, C.c_float, vp]
    lib.lm_meanpool_varlen_f16.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.lm_layer_tail_h384_f16.argtypes = [vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp, vp, i64, i32, C.c_float, vp]
    lib.lm_layer_tail_pack_h384.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    lib.lm_qkv_h384_f16.argtypes = [vp, vp, vp, i32, vp, i64, vp]
    lib.lm_qkv_attn_h384_f16.argtypes = [vp, vp, vp, vp, i32, i32, i64, vp, vp]
    lib.lm_h384_first_half_form.argtypes = [i32, i32, i64, i32]
    lib.lm_qkv_pack_h384.argtypes = [vp, i32, vp, vp]
    lib.lm_gemm_ws_h384_f16.argtypes = [vp, vp, vp, i32, vp, i64, vp]
    lib.lm_gemm_f16.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, i64, vp]
    lib.lm_pack_tokens.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    lib.lm_tokens_create.argtypes = [vp, vp, i64, C.c_int, C.POINTER(vp)]
    lib.lm_tokens_free.argtypes = [vp]
    lib.lm_tokens_free.restype = None
    lib.lm_tokens_gather.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.lm_tokens_count.argtypes = [vp]
    lib.lm_tokens_count.restype = i64
    lib.lm_bert_h384_workspace_bytes.argtypes = [i64]
    lib.lm_bert_h384_workspace_bytes.restype = C.c_size_t
    lib.lm_bert_h384_forward_packed.argtypes = [C.POINTER(BertH384), vp, vp, vp, i32, i64, i32, vp, C.c_size_t, vp, vp]
    lib.lm_bert_workspace_bytes.argtypes = [C.POINTER(Bert), i64]
    lib.lm_bert_workspace_bytes.restype = C.c_size_t
    lib.lm_bert_forward_packed.argtypes = [C.POINTER(Bert), vp, vp, vp, i32, i64, i32, vp, C.c_size_t, vp, vp]
    lib.lm_clspool_varlen_f16.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.lm_recompute_create.argtypes = [C.POINTER(BertH384), vp, i32, i64, C.POINTER(vp)]
    lib.lm_recompute_create_general.argtypes = [C.POINTER(Bert), vp, i32, i64, C.POINTER(vp)]
    lib.lm_recompute_free.argtypes = [vp]
    lib.lm_recompute_free.restype = None
    lib.lm_recompute_provider.argtypes = [vp, vp, i32, C.POINTER(vp), vp]
    lib.lm_recompute_embed.argtypes = [vp, vp, i32, vp, vp]
    lib.lm_recompute_get_stats.argtypes = [vp, C.POINTER(RecomputeStats)]
    lib.lm_index_set_recompute.argtypes = [vp, vp]
    lib.lm_kernel_timing_enable.argtypes = [C.c_uint32]
    lib.lm_kernel_timing_read.argtypes = [C.POINTER(KernelTime), i32, i32]
    _lib = lib
    return lib


def last_error() -> str:
    return (load().lm_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    """Map LM_E* codes onto the exceptions the reference raises at the same places
    (hnsw_backend.py:134,143,189-196; searcher_base.py:54)."""
    if rc == LM_OK:
        return
    msg = f"{what}: {last_error()}" if what else last_error()
    if rc == LM_EINVAL or rc == LM_EFORMAT:
        raise ValueError(msg)
    if rc == LM_ENOENT:
        raise FileNotFoundError(msg)
    raise LeannMi355xError(msg)


def kernel_timing_enable(mask: int) -> None:
    """Event pairs around the library's own launches of the encoder kernels whose bit (1 << KT_*) is set; 0 = off (lm_timing.cpp)."""
    check(load().lm_kernel_timing_enable(int(mask)), "lm_kernel_timing_enable")


def kernel_timing_read(reset: bool = False) -> dict:
    """{kernel name: {"launches", "ms", "work"}} accumulated since the last reset (waits for the pairs recorded so far)."""
    arr = (KernelTime * KT_COUNT)()
    check(load().lm_kernel_timing_read(arr, KT_COUNT, 1 if reset else 0), "lm_kernel_timing_read")
    return {a.name.decode(): {"launches": int(a.launches), "ms": float(a.ms), "work": float(a.work)} for a in arr}


def device_count() -> int:
    return int(load().lm_device_count())


def require_gpu() -> None:
    if device_count() <= 0:
        raise LeannMi355xError("no HIP device visible: leann-backend-mi355x needs an MI355X (gfx950) GPU and has no CPU fallback")
