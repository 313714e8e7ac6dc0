"""ctypes binding of libmobgt_hip.so (the C ABI declared in include/mobgt_hip.h).

There is no CPU fallback: importing this module works anywhere (so that CPU-only tests can check
the exported symbols), but every compute entry point raises if the library is missing, and the
library itself only contains gfx950 code objects.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOBGT_HIP_LIB") or os.path.join(_HERE, "libmobgt_hip.so")     # (override: A/B runs of two builds)
CSRC = os.path.join(_HERE, "csrc")

F32, BF16 = 0, 1
I64, I32, I16, U8 = 0, 1, 2, 3

_c = ctypes
_vp, _i, _i64, _f, _u64, _u32 = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_uint64, _c.c_uint32

ABI_VERSION = 3          # include/mobgt_hip.h: MOBGT_ABI_VERSION the table below was written for

SIGNATURES = {
    "mobgt_abi_version": (_i, []),
    "mobgt_build_info": (_c.c_char_p, []),
    "mobgt_attn_bias_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64,
                                 _f, _f, _u64, _vp, _i, _i, _vp]),
    "mobgt_attn_bias_bwd": (_i, [_vp] * 14 + [_i, _i, _i, _i] + [_i64] * 8 + [_f, _f, _u64, _vp, _i, _i, _i, _i, _vp]),
    "mobgt_attn_bias_bwd_fused_z": (_i, [_vp] * 14 + [_i, _i, _i, _i] + [_i64] * 8 + [_f, _f, _u64, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mobgt_dropout_keep_host": (_i, [_u64, _i, _i, _i, _i, _i, _i, _f]),
    "mobgt_attn_dropout_mask_host": (_i, [_u64, _i, _i, _i, _f, _vp]),
    "mobgt_dropout_mask_host": (_i, [_u64, _u32, _i64, _i64, _i, _f, _vp]),
    "mobgt_bias_pack": (_i, [_vp, _i, _i64, _i64, _i64, _i64, _vp, _vp, _i, _i, _i, _i, _i64, _vp]),
    "mobgt_build_bias": (_i, [_vp] * 10 + [_i] * 9 + [_i64, _i, _i, _i, _vp]),
    "mobgt_build_bias_bwd": (_i, [_vp, _i, _i, _i64] + [_vp] * 8 + [_i] * 9 + [_i64, _i, _i, _vp]),
    "mobgt_build_bias_bwd_set_workgroups": (_i, [_i]),
    "mobgt_spd_workspace_bytes": (_i64, [_i, _i]),
    "mobgt_spd_batched": (_i, [_vp] * 9 + [_i, _i, _i, _vp]),
    "mobgt_spd_set_spin_limit": (_i, [_i64]),
    "mobgt_collate_finish": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _i, _i, _vp]),
    "mobgt_floyd_warshall_workspace_bytes": (_i64, [_i]),
    "mobgt_floyd_warshall": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "mobgt_gen_edge_input": (_i, [_i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "mobgt_get_all_edges": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mobgt_dropout_add_ln_fwd": (_i, [_vp] * 9 + [_i64, _i, _f, _u64, _vp, _c.c_uint32, _i, _vp]),
    "mobgt_dropout_add_ln_bwd": (_i, [_vp] * 12 + [_i64, _i, _f, _u64, _vp, _c.c_uint32, _i, _vp]),
    "mobgt_gelu_fwd": (_i, [_vp, _vp, _i64, _i, _vp]),
    "mobgt_gelu_bwd_colsum": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "mobgt_stock_tokens_fwd": (_i, [_vp, _vp, _vp, _i, _i] + [_vp] * 5 + [_i, _i, _i, _i64, _i64, _i64, _f, _u64, _vp, _c.c_uint32, _vp]),
    "mobgt_partial_sum_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "mobgt_layer_wgrad_big": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "mobgt_layer_wgrad_big_tiles": (_i, [_i, _i]),
    "mobgt_layer_wgrad_big_splits": (_i, [_i64, _i]),
    "mobgt_stock_tail_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i] + [_vp] * 4 + [_i, _i, _i, _i64, _i64, _i64, _i64, _f, _u64, _vp, _c.c_uint32]
                             + [_vp] * 5 + [_i, _i, _i, _i] + [_vp]),
    "mobgt_stock_front_fwd": (_i, [_vp, _vp, _vp, _i, _i] + [_vp] * 5 + [_i, _i, _i, _i64, _i64, _i64, _f, _u64, _vp, _c.c_uint32]
                              + [_i, _vp, _vp, _vp, _vp, _vp] + [_i, _vp, _vp, _vp, _i, _i, _i, _i] + [_vp]),
    "mobgt_stock_tokens_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i] + [_vp] * 4 + [_i, _i, _i, _i64, _i64, _i64, _i64, _f, _u64, _vp, _c.c_uint32, _vp]),
    "mobgt_token_ln_fwd": (_i, [_vp] * 6 + [_i, _i, _i, _f, _vp]),
    "mobgt_token_ln_bwd": (_i, [_vp] * 8 + [_i, _i, _i, _vp]),
    "mobgt_cross_entropy": (_i, [_vp, _vp, _i64, _vp, _vp, _i, _i, _vp]),
    "mobgt_gradient_tail_loss": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _i64, _f, _vp]),
    "mobgt_dropout": (_i, [_vp, _vp, _i64, _i, _f, _u64, _vp, _c.c_uint32, _vp]),
    "mobgt_colsum": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "mobgt_linear_wgrad_group": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _vp]),
    "mobgt_linear_wgrad_multi": (_i, [_i] + [_vp] * 15 + [_vp]),
    "mobgt_linear_wgrad_multi_hop": (_i, [_i] + [_vp] * 15 + [_i] + [_vp] * 5 + [_i, _i, _i] + [_vp]),
    "mobgt_layer_backward_tail": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _vp, _i64, _vp, _i64, _vp, _i64,
                                       _i, _i, _i, _vp]),
    "mobgt_linear_wgrad": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _vp]),
    "mobgt_linear_wgrad_masked": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _f, _f, _f, _vp, _vp, _i64, _vp, _i, _i64, _i, _i, _vp]),
    "mobgt_linear_wgrad_bias": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i, _i, _i, _vp]),
    "mobgt_linear_wgrad_mixed": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _vp]),
    "mobgt_embed_gather_sum": (_i, [_vp, _vp, _i, _vp, _i64, _i, _i64, _i, _vp]),
    "mobgt_embed_scatter_add": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _i, _i64, _i, _vp]),
    "mobgt_embed_gather_concat": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _i64, _i, _vp]),
    "mobgt_embed_scatter_concat": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i64, _i64, _i, _vp]),
    "mobgt_embed_gather_multi": (_i, [_i] + [_vp] * 9 + [_i64, _i, _i, _vp, _i, _vp]),
    "mobgt_hop_table_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mobgt_hop_table_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mobgt_target_rank": (_i, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "mobgt_skinny_linear_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mobgt_skinny_linear_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mobgt_skinny_linear_dx": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "mobgt_skinny_linear_bwd_both": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mobgt_assemble_tokens_fwd": (_i, [_vp] * 7 + [_i, _i, _i, _f, _f, _u64, _vp, _c.c_uint32, _c.c_uint32, _c.c_uint32, _vp]),
    "mobgt_token_fwd_chain": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _f,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _u64, _vp, _u32, _u32, _u32, _vp]),
    "mobgt_assemble_tokens_qkv": (_i, [_vp] * 10 + [_i, _i, _i, _f, _f, _u64, _vp, _c.c_uint32, _c.c_uint32, _c.c_uint32, _vp]),
    "mobgt_assemble_tokens_bwd": (_i, [_vp] * 5 + [_i, _i, _i, _f, _f, _u64, _vp, _c.c_uint32, _c.c_uint32, _c.c_uint32, _vp]),
    "mobgt_bias_act_fwd": (_i, [_vp, _vp, _vp, _i64, _i, _f, _f, _u64, _vp, _c.c_uint32, _vp]),
    "mobgt_bias_act_fwd_t": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _i, _f, _f, _u64, _vp, _c.c_uint32, _vp]),
    "mobgt_bias_act_bwd": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _f, _f, _u64, _vp, _c.c_uint32, _vp]),
    "mobgt_head_act_fwd": (_i, [_vp] * 6 + [_i, _i, _f, _f, _f, _u64, _vp, _c.c_uint32, _vp]),
    "mobgt_head_act_bwd": (_i, [_vp] * 9 + [_i, _i, _f, _f, _f, _u64, _vp, _c.c_uint32, _vp]),
    "mobgt_adamw_flat": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _vp]),
    "mobgt_small_gemm_f32": (_i, [_vp, _i64, _vp, _i64, _i, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "mobgt_small_gemm_f32_act": (_i, [_vp, _i64, _vp, _f, _f, _f, _vp, _i64, _i, _vp, _i, _f, _f, _u64, _vp, _c.c_uint32, _vp, _i64, _i, _vp, _i64, _vp, _i, _i, _i, _i, _vp]),
    "mobgt_layer_gemm": (_i, [_vp, _i64, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _vp, _i, _i, _i, _vp]),
    "mobgt_ln_gemm_fwd": (_i, [_vp] * 8 + [_i64, _i, _f, _u64, _vp, _c.c_uint32, _vp, _i64, _vp, _vp, _i64, _i, _vp, _i, _vp]),
    "mobgt_ln_gemm_bwd": (_i, [_vp] * 12 + [_i64, _i, _f, _u64, _vp, _c.c_uint32, _vp, _i64, _vp, _i64, _i, _vp, _i, _vp]),
    "mobgt_mask_gemm_workspace_bytes": (_i64, [_i, _i]),
    "mobgt_mask_gemm": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _vp]),
    "mobgt_front_sgemm_job": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _f, _vp, _i64, _vp, _i64, _i, _i, _i, _i]),
    "mobgt_front_sgemm_pending": (_i, []),
    "mobgt_mask_gemm_l1_fwd": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _f, _f, _u64, _vp, _u32, _vp, _vp, _i64, _vp, _i64, _i, _i, _vp]),
    "mobgt_mask_rows_fwd": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mobgt_mask_rows_bwd_lds_bytes": (_i64, [_i64, _i]),
    "mobgt_mask_rows_bwd": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "mobgt_spmm_csr": (_i, [_vp] * 5 + [_i64, _vp, _vp, _i64, _i64, _i, _vp]),
    "mobgt_spmm_csr_t_rows": (_i, [_vp] * 5 + [_i64, _vp, _i64, _i64, _i, _vp]),
    "mobgt_spmm_csr_t_rows_gather": (_i, [_vp] * 7 + [_i64, _vp, _i64, _i64, _i64, _i, _vp]),
    "mobgt_pack_mfma_b": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mobgt_layer_chain_bwd": (_i, [_vp] * 24 + [_i64, _i, _i, _f, _u64, _vp, _c.c_uint32, _c.c_uint32, _vp, _vp, _i] + [_vp] * 9 + [_vp, _vp]),
    "mobgt_layer_chain_bwd_big": (_i, [_vp] * 25 + [_i64, _i, _i, _f, _u64, _vp, _c.c_uint32, _c.c_uint32, _vp, _vp, _vp]),
    "mobgt_layer_chain_bwd_preln": (_i, [_vp] * 24 + [_i64, _i, _i, _f, _u64, _vp, _c.c_uint32, _c.c_uint32, _vp, _vp, _i] + [_vp] * 9 + [_vp, _vp]),
    "mobgt_layer_chain_fwd": (_i, [_vp] * 26 + [_i64, _i, _i, _f, _u64, _vp, _c.c_uint32, _c.c_uint32, _vp, _vp]),
    "mobgt_chain_ws_bytes": (_i64, []),
    "mobgt_chain_ws_fault_offset": (_i64, []),
    "mobgt_chain_ws_limit_offset": (_i64, []),
    "mobgt_head_chain_ws_fault_offset": (_i64, []),
    "mobgt_head_chain_ws_limit_offset": (_i64, []),
    "mobgt_small_gcn_faults": (_i, [_i, _vp]),
    "mobgt_small_gcn_set_wait_limit": (_i, [_i64]),
    "mobgt_debug_occupy": (_i, [_i, _i, _i, _i64, _vp]),
    "mobgt_small_gcn_fwd": (_i, [_vp] * 14 + [_i] * 5 + [_f, _f, _u64, _vp, _c.c_uint32, _vp]),
    "mobgt_small_gcn_fwd_pack": (_i, [_vp] * 14 + [_i] * 5 + [_f, _f, _u64, _vp, _c.c_uint32, _i, _vp, _vp, _vp, _vp, _vp]
                                 + [_i, _vp, _i, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i]
                                 + [_i, _vp, _vp, _vp, _i, _i, _i, _i] + [_vp]),
    "mobgt_small_gcn_bwd": (_i, [_vp] * 18 + [_i] * 5 + [_f, _f, _u64, _vp, _c.c_uint32, _vp]),
    "mobgt_small_gcn_bwd_bias": (_i, [_vp] * 18 + [_i] * 5 + [_f, _f, _u64, _vp, _c.c_uint32]
                                 + [_i] + [_vp, _i, _i, _i64] + [_vp] * 8 + [_i] * 9 + [_i64, _i, _i, _vp]),
    "mobgt_step_prologue": (_i, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "mobgt_step_prologue_skip": (_i, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp]),
    "mobgt_head_chain_fwd": (_i, [_vp, _vp, _i, _i64, _vp, _i64] + [_vp] * 9 + [_i, _i, _i, _i, _f, _f, _f, _u64, _vp, _c.c_uint32, _vp, _vp]),
    "mobgt_head_chain_ws_bytes": (_i64, []),
    "mobgt_head_chain_bwd": (_i, [_vp] * 5 + [_i, _i64, _i64] + [_vp] * 8 + [_i, _i, _i, _i, _f, _f, _f, _u64, _vp, _c.c_uint32, _vp]),
    "mobgt_token_bwd_chain": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _u64, _vp,
                                   _c.c_uint32, _c.c_uint32, _c.c_uint32, _vp]),
    "mobgt_skinny_linear_fwd_mfma": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mobgt_skinny_linear_gtl": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "mobgt_head_input_fwd": (_i, [_vp, _vp, _i, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _vp]),
    "mobgt_head_input_bwd": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "mobgt_gather_rows_t": (_i, [_vp, _i64, _vp, _vp, _vp, _i, _i, _vp]),
    "mobgt_node_index": (_i, [_vp, _i, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
}

_lib = None


def build(force=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "mobgt_hip.h"))
    stale = force or not os.path.exists(LIB_PATH) or \
        any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-j4", "-C", CSRC])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MobGT hot path has no CPU fallback. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc).")
        # torch first: the library must bind to the HIP runtime torch has loaded (its own libamdhip64).  Loaded
        # before torch it pulls in /opt/rocm's copy, and the process then holds two runtimes -- kernels registered
        # with one, torch's streams and buffers owned by the other (every launch fails with hipErrorNoDevice).
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        handle.mobgt_abi_version.restype = _i
        have = handle.mobgt_abi_version()
        if have != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI version {have}, this binding was written for {ABI_VERSION}: a stale build "
                               "(arguments would be shifted silently) -- rebuild with __graft_entry__.build()")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class MobgtError(RuntimeError):
    pass


_ERR = {-1: "unsupported dimension (MOBGT_EBADDIM)", -2: "alignment/stride violation (MOBGT_EALIGN)",
        -3: "unknown dtype code (MOBGT_EDTYPE)"}


def check(rc, what):
    if rc != 0:
        raise MobgtError(f"{what} failed: {_ERR.get(rc, f'hipError_t {rc}')}")
