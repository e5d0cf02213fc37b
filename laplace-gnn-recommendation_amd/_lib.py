"""ctypes binding of liblaplace_hip.so (include/laplace_hip.h).  Fails loudly, never falls back."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# LAPLACE_HIP_LIB names a VARIANT build of the same library (A/B and timing probes under tools/): the product library in
# the package directory is never overwritten by an experiment.  It is still this library or nothing — no fallback.
LIB_PATH = os.environ.get("LAPLACE_HIP_LIB") or os.path.join(PKG_DIR, "liblaplace_hip.so")

MI_ABI_VERSION = 9
MI_SPMM_GROUP = 32


class MiError(RuntimeError):
    pass


class SpmmPlanStruct(Structure):
    _fields_ = [
        ("chunk", c_int32), ("n_long_rows", c_int32), ("n_items", c_int32), ("n_launch", c_int32),
        ("long_rows", c_void_p), ("item_ptr", c_void_p), ("items", c_void_p), ("long_index", c_void_p),
        ("band", c_int32), ("n_bands", c_int32),
        ("epos", c_void_p), ("ecol", c_void_p), ("eval", c_void_p),
    ]


class SpmmPlanInfo(Structure):
    _fields_ = [("n_long_rows", c_int64), ("n_items", c_int64), ("n_launch", c_int64), ("nnz_long", c_int64),
                ("chunk", c_int32), ("band", c_int32), ("n_bands", c_int32), ("queue_len", c_int32),
                ("queue_start", c_int32 * 9), ("keys_in_second", c_int32)]


class AdamArgs(Structure):
    _fields_ = [("p", c_void_p), ("ldp", c_int64), ("m", c_void_p), ("v", c_void_p), ("reg_w", c_void_p),
                ("lr", c_double), ("beta1", c_double), ("beta2", c_double), ("eps", c_double), ("step", c_int64)]


class SpmmSweepStruct(Structure):
    _fields_ = [("col", c_void_p), ("val", c_void_p), ("stream_ptr", c_void_p), ("slot_of", c_void_p),
                ("n_streams", c_int32), ("n_slots", c_int32), ("progress", c_void_p), ("epoch", c_int32), ("slack", c_int32)]


class SpmmExStruct(Structure):
    _fields_ = [("x_map", c_void_p), ("addend_map", c_void_p), ("row_list", c_void_p), ("n_list_dev", c_void_p),
                ("n_list", c_int64), ("adam", POINTER(AdamArgs)), ("parts", c_int32), ("hot_rows", c_int32),
                ("sweep", POINTER(SpmmSweepStruct)), ("hot_base", c_int32), ("hot_threads", c_int32),
                ("x_bits", c_void_p)]


MI_SPMM_SHORT_ROWS, MI_SPMM_SPLIT_ROWS = 1, 2
MI_ERR_UNSUPPORTED = -4
MI_ERR_WORKSPACE = -3
MI_TOPK_ITEMS_PREPARED = 1


class GemmProblem(Structure):
    _fields_ = [("trans_a", c_int32), ("trans_b", c_int32), ("m", c_int64), ("n", c_int64), ("k", c_int64),
                ("A", c_void_p), ("lda", c_int64), ("B", c_void_p), ("ldb", c_int64),
                ("k2", c_int64), ("A2", c_void_p), ("lda2", c_int64), ("B2", c_void_p), ("ldb2", c_int64),
                ("a_mask", c_void_p), ("bias", c_void_p), ("C", c_void_p), ("ldc", c_int64),
                ("accumulate", c_int32), ("act", c_int32)]


class SamplerDesc(Structure):
    _fields_ = [("batch", c_int32), ("n_hops", c_int32), ("num_neighbors", c_int32), ("k", c_int32),
                ("randomization", c_int32), ("max_pos", c_int32), ("max_neg", c_int32), ("reserved", c_int32),
                ("num_users", c_int64), ("num_articles", c_int64), ("num_edges", c_int64), ("id_max", c_int64),
                ("users_ptr", c_void_p), ("users_idx", c_void_p), ("articles_ptr", c_void_p), ("articles_idx", c_void_p),
                ("positive_edges_ratio", c_double), ("negative_edges_ratio", c_double), ("reject_min_entries", c_int64),
                ("cand_ptr", c_void_p), ("cand_idx", c_void_p)]


MI_RANKER_MAX_LAYERS, MI_RANKER_MAX_COLS, MI_RANKER_MAX_PARAMS = 4, 16, 48


class RankerConv(Structure):
    _fields_ = [("w_l", c_void_p), ("b_l", c_void_p), ("w_r", c_void_p), ("gw_l", c_void_p), ("gb_l", c_void_p), ("gw_r", c_void_p),
                ("c_src", c_int32), ("c_dst", c_int32), ("c_out", c_int32), ("reserved", c_int32)]


class RankerNorm(Structure):
    _fields_ = [("gamma", c_void_p), ("beta", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p),
                ("num_batches_tracked", c_void_p), ("g_gamma", c_void_p), ("g_beta", c_void_p), ("momentum", c_float), ("eps", c_float)]


class RankerLinear(Structure):
    _fields_ = [("w", c_void_p), ("b", c_void_p), ("gw", c_void_p), ("gb", c_void_p), ("in_", c_int32), ("out", c_int32)]


class RankerParam(Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n", c_int64)]


class RankerModel(Structure):
    _fields_ = [("n_enc_layers", c_int32), ("n_dec_layers", c_int32), ("aggr", c_int32), ("batch_normalize", c_int32),
                ("p_dropout", c_float), ("max_norm", c_float), ("n_cols", c_int32 * 2),
                ("tables", (c_void_p * MI_RANKER_MAX_COLS) * 2), ("table_rows", (c_int64 * MI_RANKER_MAX_COLS) * 2),
                ("dims", (c_int32 * MI_RANKER_MAX_COLS) * 2), ("conv", (RankerConv * 2) * MI_RANKER_MAX_LAYERS),
                ("norm", RankerNorm * 2), ("dec", RankerLinear * MI_RANKER_MAX_LAYERS), ("n_params", c_int32), ("apply_adam", c_int32),
                ("params", RankerParam * MI_RANKER_MAX_PARAMS), ("lr", c_double), ("beta1", c_double), ("beta2", c_double),
                ("eps", c_double), ("step", c_int64), ("ones4", c_void_p), ("n_ones", c_int64)]


class RankerBatch(Structure):
    _fields_ = [("n_nodes", c_int64 * 2), ("x", c_void_p * 2), ("by_customer_ptr", c_void_p), ("by_customer_col", c_void_p),
                ("by_article_ptr", c_void_p), ("by_article_col", c_void_p), ("nnz", c_int64), ("n_label", c_int64),
                ("label_row", c_void_p), ("label_col", c_void_p), ("label", c_void_p), ("seed", c_uint64), ("step", c_uint64),
                ("loss", c_void_p), ("label_f32", c_void_p), ("aux_stream", c_void_p), ("ev_fork", c_void_p), ("ev_join", c_void_p),
                ("logits", c_void_p)]


MI_PINSAGE_MAX_LAYERS = 4


class PinsageBatchDesc(Structure):
    _fields_ = [("batch", c_int64), ("n_items", c_int64), ("iu_ptr", c_void_p), ("iu_idx", c_void_p), ("ui_ptr", c_void_p),
                ("ui_idx", c_void_p), ("walk_length", c_int32), ("num_walks", c_int32), ("num_neighbors", c_int32),
                ("num_layers", c_int32), ("restart_prob", c_double), ("pos_scratch", c_void_p)]


class PinsageBlockOut(Structure):
    _fields_ = [("src_ids", c_void_p), ("edge_src", c_void_p), ("edge_dst", c_void_p), ("weights", c_void_p),
                ("dst_rowptr", c_void_p), ("dst_col", c_void_p), ("dst_val", c_void_p),
                ("src_rowptr", c_void_p), ("src_col", c_void_p), ("src_val", c_void_p)]


class PinsageBatchOut(Structure):
    _fields_ = [("seeds", c_void_p), ("pos_u", c_void_p), ("pos_v", c_void_p), ("neg_v", c_void_p), ("counts", c_void_p),
                ("blocks", PinsageBlockOut * MI_PINSAGE_MAX_LAYERS)]


class WgradProblem(Structure):
    _fields_ = [("k", c_int64), ("m", c_int32), ("n1", c_int32), ("n2", c_int32), ("reserved", c_int32),
                ("dy", c_void_p), ("mask", c_void_p), ("b1", c_void_p), ("b2", c_void_p),
                ("gw1", c_void_p), ("gb", c_void_p), ("gw2", c_void_p)]


MI_PINSAGE_MAX_PARAMS = 24


class PinsageConv(Structure):
    _fields_ = [("q_w", c_void_p), ("q_b", c_void_p), ("w_w", c_void_p), ("w_b", c_void_p),
                ("g_q_w", c_void_p), ("g_q_b", c_void_p), ("g_w_w", c_void_p), ("g_w_b", c_void_p)]


class PinsageModel(Structure):
    _fields_ = [("n_layers", c_int32), ("hidden", c_int32), ("n_items", c_int64),
                ("proj", c_void_p), ("g_proj", c_void_p), ("m_proj", c_void_p), ("v_proj", c_void_p),
                ("bias", c_void_p), ("g_bias", c_void_p), ("conv", PinsageConv * MI_PINSAGE_MAX_LAYERS),
                ("params", RankerParam * MI_PINSAGE_MAX_PARAMS), ("n_params", c_int32), ("apply_adam", c_int32),
                ("p_dropout", c_float), ("reserved", c_int32), ("lr", c_double), ("beta1", c_double), ("beta2", c_double),
                ("eps", c_double), ("step", c_int64), ("ones4", c_void_p), ("n_ones", c_int64)]


class PinsageStepBlock(Structure):
    _fields_ = [("n_src", c_int64), ("n_dst", c_int64), ("nnz", c_int64), ("src_ids", c_void_p),
                ("dst_rowptr", c_void_p), ("dst_col", c_void_p), ("dst_val", c_void_p),
                ("src_rowptr", c_void_p), ("src_col", c_void_p), ("src_val", c_void_p)]


class PinsageStepBatch(Structure):
    _fields_ = [("n_blocks", c_int32), ("reserved", c_int32), ("blocks", PinsageStepBlock * MI_PINSAGE_MAX_LAYERS),
                ("n_seeds", c_int64), ("n_pairs", c_int64), ("seeds", c_void_p), ("pos_u", c_void_p), ("pos_v", c_void_p),
                ("neg_v", c_void_p), ("seed", c_uint64), ("step", c_uint64), ("loss", c_void_p),
                ("rows_out", c_void_p), ("bias_out", c_void_p)]


class PinsageGradList(Structure):
    _fields_ = [("n_rows", c_int64), ("n_seeds", c_int64), ("ids", c_void_p), ("rows", c_void_p), ("bias", c_void_p)]


P = c_void_p
_PROTOTYPES = {
    # name: (restype, [argtypes])
    "mi_abi_version": (c_int32, []),
    "mi_error_string": (c_char_p, [c_int32]),
    "mi_coo_to_csr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "mi_coo_to_csr_i32": (c_int32, [c_int64, c_int64, c_int64, P, P, P, P, P, P, c_size_t, P]),
    "mi_csr_transpose_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "mi_csr_transpose_i32": (c_int32, [c_int64, c_int64, c_int64, P, P, P, P, P, P, c_size_t, P]),
    "mi_gather_f32": (c_int32, [c_int64, P, P, P, P]),
    "mi_gcn_norm_csr_f32": (c_int32, [c_int64, c_int64, P, P, P, P, P, P]),
    "mi_scale_csr_f32": (c_int32, [c_int64, c_int64, P, P, P, P, P, P, P]),
    "mi_spmm_plan_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "mi_spmm_plan_count": (c_int32, [c_int64, c_int64, P, P, c_int32, c_int32, P, c_size_t, POINTER(SpmmPlanInfo), P]),
    "mi_spmm_plan_count_range": (c_int32, [c_int64, c_int64, P, P, c_int32, c_int32, c_int32, P, c_size_t, POINTER(SpmmPlanInfo), P]),
    "mi_spmm_plan_fill": (c_int32, [c_int64, P, POINTER(SpmmPlanInfo), POINTER(SpmmPlanStruct), P, c_size_t, P]),
    "mi_spmm_workspace_bytes": (c_size_t, [POINTER(SpmmPlanStruct), c_int64]),
    "mi_spmm_plan_pack_workspace_bytes": (c_size_t, [c_int64]),
    "mi_spmm_plan_pack_entries": (c_int32, [POINTER(SpmmPlanStruct), c_int64, c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "mi_map_live_bits_i32": (c_int32, [c_int64, c_void_p, c_void_p, c_void_p]),
    "mi_spmm_csr_f32": (c_int32, [c_int64, c_int64, P, P, P, P, c_int64, P, c_int64, P, c_int64, P, c_int64,
                                  c_float, POINTER(SpmmPlanStruct), P, c_size_t, P]),
    "mi_spmm_csr_ex_f32": (c_int32, [c_int64, c_int64, P, P, P, P, c_int64, P, c_int64, P, c_int64, P, c_int64,
                                     c_float, POINTER(SpmmPlanStruct), POINTER(SpmmExStruct), P, c_size_t, P]),
    "mi_gather_rows_f32": (c_int32, [c_int64, P, P, c_int64, P, c_int64, P, c_int64, P, c_int64, c_int32, c_float, P]),
    "mi_scatter_rows_f32": (c_int32, [c_int64, P, P, c_int64, P, c_int64, P, c_int64, P, c_int64, P]),
    "mi_batch_nodes_workspace_bytes": (c_size_t, [c_int64]),
    "mi_batch_nodes_i32": (c_int32, [c_int64, c_int64, c_int64, P, P, P, P, P, P, P, c_size_t, P]),
    "mi_csr_expand_rows": (c_int32, [c_int64, P, P, c_int64, P]),
    "mi_sample_bpr_batch": (c_int32, [c_int64, c_int64, P, P, P, c_int64, c_int32, c_int32, c_uint64, c_uint64,
                                      P, P, P, P]),
    "mi_bpr_workspace_bytes": (c_size_t, [c_int64]),
    "mi_bpr_fwd_bwd_f32": (c_int32, [c_int64, c_int64, c_int64, P, P, P, P, c_int64, P, c_int64,
                                     c_float, c_float, c_float, P, P, c_int64, P, P, P, c_size_t, P]),
    "mi_gemm_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "mi_gemm_f32": (c_int32, [c_int32, c_int32, c_int64, c_int64, c_int64, P, c_int64, P, c_int64, P, P, c_int64,
                              c_int32, c_int32, P, c_size_t, P]),
    "mi_gemm_group_workspace_bytes": (c_size_t, [POINTER(GemmProblem), c_int32]),
    "mi_gemm_group_f32": (c_int32, [POINTER(GemmProblem), c_int32, P, c_size_t, P]),
    "mi_topk_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "mi_topk_excl_f32": (c_int32, [c_int64, c_int64, c_int64, c_int64, P, P, c_int64, P, c_int64, P, P, P, P, P,
                                   c_size_t, P]),
    "mi_topk_excl_ex_f32": (c_int32, [c_int64, c_int64, c_int64, c_int64, P, P, c_int64, P, c_int64, P, P, P, P, P,
                                      c_size_t, c_uint32, P]),
    "mi_topk_prefilter_scores_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "mi_topk_prefilter_scores_f32": (c_int32, [c_int64, c_int64, c_int64, P, P, c_int64, P, c_int64, P, P, P, c_size_t, P]),
    "mi_segment_max_f32": (c_int32, [c_int64, c_int64, P, P, P, c_int64, P, c_int64, P, P]),
    "mi_segment_max_bwd_f32": (c_int32, [c_int64, c_int64, P, P, P, P, c_int64, P, c_int64, P]),
    "mi_embed_concat_f32": (c_int32, [c_int64, c_int32, P, POINTER(c_void_p), POINTER(c_int64), POINTER(c_int32),
                                      c_float, P, c_int64, P]),
    "mi_batchnorm_workspace_bytes": (c_size_t, [c_int64]),
    "mi_batchnorm_fwd_f32": (c_int32, [c_int64, c_int64, P, c_int64, P, P, P, P, c_float, c_float, c_int32, P, P, P, c_int64,
                                       P, c_size_t, P]),
    "mi_batchnorm_bwd_f32": (c_int32, [c_int64, c_int64, P, c_int64, P, c_int64, P, P, P, P, c_int64, P, P, P, c_size_t, P]),
    "mi_bce_logits_f32": (c_int32, [c_int64, P, P, P, P, P]),
    "mi_gather_cat_f32": (c_int32, [c_int64, c_int64, c_int64, P, P, P, c_int64, P, c_int64, P, c_int64, P]),
    "mi_gather_cat_bwd_max_edges": (c_int64, []),
    "mi_gather_cat_bwd_f32": (c_int32, [c_int64, c_int64, c_int64, P, P, c_int64, P, c_int64, P]),
    "mi_match_common_items_i32": (c_int32, [c_int64, P, P, P, P, P, c_int32, P, P, P]),
    "mi_gemm_group_supported": (c_int32, [P, c_int32]),
    "mi_sage_wgrad_supported": (c_int32, [POINTER(WgradProblem), c_int32]),
    "mi_sage_wgrad_workspace_bytes": (c_size_t, [POINTER(WgradProblem), c_int32]),
    "mi_sage_wgrad_f32": (c_int32, [POINTER(WgradProblem), c_int32, P, c_size_t, P]),
    "mi_linear1_bwd_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "mi_linear1_bwd_f32": (c_int32, [c_int64, c_int64, P, P, P, c_int64, P, c_int64, P, P, P, c_size_t, P]),
    "mi_ranker_sizeof": (c_int64, [c_int32]),
    "mi_ranker_step_workspace_bytes": (c_size_t, [POINTER(RankerModel), POINTER(RankerBatch)]),
    "mi_ranker_step_f32": (c_int32, [POINTER(RankerModel), POINTER(RankerBatch), P, c_size_t, P]),
    "mi_ranker_step_check": (c_int32, [POINTER(RankerModel), POINTER(RankerBatch), P, c_size_t]),
    "mi_ranker_adam_f32": (c_int32, [POINTER(RankerModel), ctypes.c_float, P]),
    "mi_match_same_location_i32": (c_int32, [c_int64, P, P, P, P, P, P, c_int32, P, P, P]),
    "mi_sampler_workspace_bytes": (c_size_t, [POINTER(SamplerDesc)]),
    "mi_sampler_count": (c_int32, [POINTER(SamplerDesc), P, c_uint64, c_uint64, P, c_size_t, POINTER(c_int64), P]),
    "mi_sampler_count_async": (c_int32, [POINTER(SamplerDesc), P, c_uint64, c_uint64, P, c_size_t, P, P]),
    "mi_sampler_emit": (c_int32, [POINTER(SamplerDesc), P, P, c_size_t, POINTER(c_int64), P, P, P, P, P, P, P, P]),
    "mi_sampler_emit3": (c_int32, [POINTER(SamplerDesc), P, P, c_size_t, POINTER(c_int64), P, P, P, P, P, P, P, c_int32, P]),
    "mi_sampler_emit_csr": (c_int32, [POINTER(SamplerDesc), P, c_size_t, POINTER(c_int64), P, P, P, P, P, P]),
    "mi_pinsage_item_pairs": (c_int32, [c_int64, c_int64, P, P, P, P, c_uint64, c_uint64, P, P, P, P]),
    "mi_pinsage_neighbors_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "mi_pinsage_neighbors": (c_int32, [c_int64, P, P, P, P, P, c_int32, c_double, c_int32, c_int32, c_int32,
                                       c_uint64, c_uint64, P, P, P, c_size_t, P]),
    "mi_pinsage_batch_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32, c_int32, c_int32]),
    "mi_pinsage_sample_batch": (c_int32, [POINTER(PinsageBatchDesc), c_uint64, c_uint64, POINTER(PinsageBatchOut), P, c_size_t, P]),
    "mi_pinsage_step_sizeof": (c_int64, [c_int32]),
    "mi_pinsage_step_workspace_bytes": (c_size_t, [POINTER(PinsageModel), POINTER(PinsageStepBatch)]),
    "mi_pinsage_step_f32": (c_int32, [POINTER(PinsageModel), POINTER(PinsageStepBatch), P, c_size_t, P]),
    "mi_pinsage_step_check": (c_int32, [POINTER(PinsageModel), POINTER(PinsageStepBatch), P, c_size_t]),
    "mi_pinsage_apply_f32": (c_int32, [POINTER(PinsageModel), POINTER(PinsageGradList), c_int32, ctypes.c_float, P]),
    "mi_adam_dense_f32": (c_int32, [c_int64, c_int64, P, c_int64, P, c_int64, P, P, P,
                                    c_double, c_double, c_double, c_double, c_int64, P]),
}

_LIB = None


def lib() -> ctypes.CDLL:
    """Load (once) and return the shared library; raises MiError if it is not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MiError(
                f"{LIB_PATH} is missing: build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # torch first: the process must hold ONE HIP runtime, the one torch ships; loaded after ours, torch's copy
        # would be a second runtime that sees no device ("no ROCm-capable device is detected" from our calls)
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        got = handle.mi_abi_version()
        if got != MI_ABI_VERSION:
            raise MiError(f"ABI mismatch: library {got}, binding {MI_ABI_VERSION}")
        _LIB = handle
    return _LIB


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mi_error_string(int(rc))
        raise MiError(f"{what} failed with code {rc}: {msg.decode() if msg else '?'}")


def exported_symbols() -> list[str]:
    return sorted(_PROTOTYPES)


_raw_stream = None
_cur_device = None


def current_stream() -> int:
    """The current HIP stream of the current device as a raw handle.  torch.cuda.current_stream() builds a Stream object
    through several Python layers (~10 us per call, measured: 0.2 ms of a 2 ms ranker iteration); torch's raw getter
    is ~0.3 us."""
    global _raw_stream, _cur_device
    if _raw_stream is None:
        import torch as t
        _raw_stream = getattr(t._C, "_cuda_getCurrentRawStream", False)
        _cur_device = getattr(t._C, "_cuda_getDevice", False)
    if _raw_stream and _cur_device:
        return _raw_stream(_cur_device())
    import torch as t
    return t.cuda.current_stream().cuda_stream
