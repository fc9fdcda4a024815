"""ctypes binding of libldmk.so (include/ldmk.h).  No CPU fallback: if the library is missing or a
call fails, this raises -- the product path is the HIP path or nothing."""
import ctypes as C
import os

from . import switches

_HERE = os.path.dirname(os.path.abspath(__file__))
# LDMK_LIBRARY: another build of the same library (A/B timing of a kernel change on one box); still no fallback
LIB_PATH = switches.get("LDMK_LIBRARY") or os.path.join(_HERE, "libldmk.so")

A_ROWS, A_CONV3X3 = 0, 1
TF_NONE, TF_AFFINE, TF_AFFINE_SILU, TF_LAYERNORM, TF_LAYERNORM_FOLDED = 0, 1, 2, 3, 4
EPI_NONE, EPI_GEGLU = 0, 1
COMPUTE_F32, COMPUTE_BF16, COMPUTE_BF16X3, COMPUTE_F16X2 = 0, 1, 2, 3
POST_NONE, POST_GROUPNORM, POST_LAYERNORM = 0, 1, 2

_fp = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class IgemmArgs(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("a0", _fp), ("a1", _fp), ("c0", C.c_int), ("c1", C.c_int),
        ("a_mode", C.c_int), ("in_h", C.c_int), ("in_w", C.c_int), ("out_h", C.c_int), ("out_w", C.c_int),
        ("stride", C.c_int), ("pad_lo", C.c_int), ("upsample", C.c_int),
        ("a_tf", C.c_int), ("tf_coef", _fp), ("row_stats", _fp), ("ln_gamma", _fp), ("ln_beta", _fp),
        ("rows_per_sample", C.c_int),
        ("w", _fp), ("b_trans", C.c_int), ("ldb", C.c_int),
        ("bias", _fp), ("batch_vec", _fp), ("batch_vec_ld", C.c_int),
        ("residual", _fp), ("epi", C.c_int),
        ("out", _fp), ("ldc", C.c_int), ("batch", C.c_int),
        ("a_bstride", C.c_longlong), ("w_bstride", C.c_longlong), ("out_bstride", C.c_longlong),
        ("alpha", C.c_float), ("tile_cfg", C.c_int), ("splitk", C.c_int),
        ("splitk_ws", _fp), ("splitk_ws_elems", C.c_longlong), ("stats_out", _fp), ("compute", C.c_int), ("splitk_counters", _fp), ("splitk_counters_len", C.c_int),
        ("w_frag", _fp),
        ("ln_colsum", _fp),
        ("raw_slabs", C.c_int),
        ("skip_a0", _fp), ("skip_a1", _fp), ("skip_c0", C.c_int), ("skip_c1", C.c_int),
        ("w_split", _fp), ("w_split_ld", C.c_int), ("w_split_bstride", C.c_longlong),
        ("a_split", _fp), ("a_split_ld", C.c_int),
        ("a_ps", _fp), ("a_ps_bstride", C.c_longlong), ("w_ps", _fp), ("w_ps_bstride", C.c_longlong), ("out_ps", _fp),
        ("w_scale_exp", C.c_int), ("range_flag", _fp),
        ("attn_kv_out", _fp), ("attn_tokens", C.c_int), ("attn_heads", C.c_int),
    ]


class PostArgs(C.Structure):
    _fields_ = [
        ("src", _fp), ("nslab", C.c_int), ("slab_stride", C.c_longlong), ("M", C.c_int), ("N", C.c_int),
        ("rows_per_sample", C.c_int), ("alpha", C.c_float), ("bias", _fp), ("batch_vec", _fp), ("batch_vec_ld", C.c_int),
        ("residual", _fp), ("ldr", C.c_int), ("geglu", C.c_int), ("raw_out", _fp), ("ld_raw", C.c_int), ("norm", C.c_int),
        ("x1", _fp), ("c1", C.c_int), ("groups", C.c_int), ("eps", C.c_float), ("gamma", _fp), ("beta", _fp),
        ("silu", C.c_int), ("norm_out", _fp), ("ld_norm", C.c_int), ("gn_cache_floats", C.c_int),
        ("gn_scratch", _fp), ("gn_scratch_elems", C.c_longlong),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("R", C.c_int), ("Kw", C.c_int), ("N", C.c_int), ("a", _fp), ("c", C.c_int), ("lda", C.c_int),
        ("a_mode", C.c_int), ("in_h", C.c_int), ("in_w", C.c_int), ("out_h", C.c_int), ("out_w", C.c_int),
        ("stride", C.c_int), ("pad_lo", C.c_int), ("upsample", C.c_int),
        ("dy", _fp), ("ldy", C.c_int), ("dw", _fp), ("ldw", C.c_int), ("accumulate", C.c_int), ("alpha", C.c_float),
        ("batch", C.c_int), ("a_bstride", C.c_longlong), ("dy_bstride", C.c_longlong), ("dw_bstride", C.c_longlong),
        ("splitr", C.c_int), ("ws", _fp), ("ws_elems", C.c_longlong), ("dbias", _fp), ("compute", C.c_int),
    ]


_SIGS = {
    "ldmk_version": (C.c_int, []),
    "ldmk_last_error": (C.c_char_p, []),
    "ldmk_init": (C.c_int, [C.c_int]),
    "ldmk_igemm": (C.c_int, [C.POINTER(IgemmArgs), _fp]),
    "ldmk_igemm_workspace_elems": (C.c_longlong, [C.POINTER(IgemmArgs)]),
    "ldmk_igemm_check": (C.c_int, [C.POINTER(IgemmArgs)]),
    "ldmk_igemm_plan": (C.c_int, [C.POINTER(IgemmArgs), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ldmk_igemm_force_config": (None, [C.c_int]),
    "ldmk_wfrag_elems": (C.c_longlong, [C.c_int, C.c_int]),
    "ldmk_pack_wfrag": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ldmk_pack_wsplit": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, _fp, C.c_int, _fp]),
    "ldmk_pack_wsplit_h2": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, _fp, C.c_int, _fp]),
    "ldmk_pack_wbf16t": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp]),
    "ldmk_winograd_tiles": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "ldmk_winograd_input": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ldmk_winograd_input_ps": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ldmk_winograd_output": (C.c_int, [_fp, _fp, _fp, C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_upconv_gather": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ldmk_upconv_gather_ps": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ldmk_upconv_gather_ps_h2": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]),
    "ldmk_winograd_input_ps_h2": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]),
    "ldmk_upconv_scatter": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_fold_layernorm": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "ldmk_attn_force_qt": (None, [C.c_int]),
    "ldmk_gn_chunks": (C.c_int, [C.c_int]),
    "ldmk_gn_partial": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ldmk_gn_finalize": (C.c_int, [_fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp, _fp, _fp]),
    "ldmk_gn_coef": (C.c_int, [_fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp, _fp, _fp, _fp]),
    "ldmk_ln_stats": (C.c_int, [_fp, C.c_int, C.c_int, C.c_float, _fp, _fp]),
    "ldmk_ln_stats_guard": (C.c_int, [_fp, C.c_int, C.c_int, C.c_float, _fp, C.c_float, _fp, _fp]),
    "ldmk_ps_bytes": (C.c_longlong, [C.c_int, C.c_int]),
    "ldmk_pack_ps": (C.c_int, [_fp, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.c_longlong, _fp, _fp]),
    "ldmk_ps_bytes_h2": (C.c_longlong, [C.c_int, C.c_int]),
    "ldmk_pack_ps_h2": (C.c_int, [_fp, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.c_longlong, C.c_int, _fp, _fp, _fp]),
    "ldmk_ln_stats_ps_h2": (C.c_int, [_fp, C.c_int, C.c_int, C.c_float, _fp, _fp, C.c_float, _fp, _fp, _fp]),
    "ldmk_ln_stats_ps": (C.c_int, [_fp, C.c_int, C.c_int, C.c_float, _fp, _fp, C.c_float, _fp, _fp]),
    "ldmk_ln_stats_split": (C.c_int, [_fp, C.c_int, C.c_int, C.c_float, _fp, _fp, C.c_int, _fp]),
    "ldmk_gn_apply": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_gn_coef_film": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_gn_apply_ps_h2": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ldmk_post": (C.c_int, [C.POINTER(PostArgs), _fp]),
    "ldmk_post_scratch_elems": (C.c_longlong, [C.POINTER(PostArgs)]),
    "ldmk_attn_self": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_attn_self_x3": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_attn_kv_split_bytes": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "ldmk_attn_self_x3p": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_attn_self_x3p_ps": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_attn_kv_split_h2_bytes": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "ldmk_attn_self_h2": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_attn_self_h2_ps": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_attn_self_h2_tiles": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_attn_self_small": (C.c_int, [_fp, C.c_int, C.c_longlong, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_attn_cross": (C.c_int, [_fp, C.c_int, _fp, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_float, _fp]),
    "ldmk_attn_cross_d": (C.c_int, [_fp, C.c_int, _fp, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_heads_gather": (C.c_int, [_fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_heads_scatter": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_softmax_rows": (C.c_int, [_fp, C.c_longlong, C.c_int, C.c_float, _fp]),
    "ldmk_dense_small": (C.c_int, [_fp, C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_timestep_embedding": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, _fp]),
    "ldmk_conv3x3_in": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_conv3x3_out": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_conv3x3_out_small": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_conv1x1_nchw": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_ddim_step": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_float, C.c_int, _fp, _fp, C.c_longlong, C.c_int, _fp, _fp,
                                 C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_advance_timestep": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_ddpm_step": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_longlong, C.c_int, _fp]),
    "ldmk_vq_nearest": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_permute3": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_audio_attention": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]),
    "ldmk_mask_rows": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_pack_conv3x3": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp]),
    "ldmk_postprocess_frames": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_add_rowvec": (C.c_int, [_fp, _fp, C.c_int, C.c_longlong, C.c_int, C.c_int, _fp]),
    # ---- training step (N1)
    "ldmk_wgrad": (C.c_int, [C.POINTER(WgradArgs), _fp]),
    "ldmk_wgrad_workspace_elems": (C.c_longlong, [C.POINTER(WgradArgs)]),
    "ldmk_wgrad_plan": (C.c_int, [C.POINTER(WgradArgs), C.POINTER(C.c_int)]),
    "ldmk_pack_dgrad3x3": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp]),
    "ldmk_gn_group_stats": (C.c_int, [_fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp]),
    "ldmk_gn_bwd_chunks": (C.c_int, [C.c_int]),
    "ldmk_gn_bwd_scratch_elems": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldmk_gn_bwd": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp,
                              C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, _fp, _fp]),
    "ldmk_ln_apply": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp]),
    "ldmk_ln_bwd_blocks": (C.c_int, [C.c_int]),
    "ldmk_ln_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_int, _fp, _fp]),
    "ldmk_geglu_fwd": (C.c_int, [_fp, _fp, C.c_longlong, C.c_int, _fp]),
    "ldmk_geglu_bwd": (C.c_int, [_fp, _fp, _fp, C.c_longlong, C.c_int, _fp]),
    "ldmk_softmax_bwd_rows": (C.c_int, [_fp, _fp, C.c_longlong, C.c_int, C.c_float, _fp]),
    "ldmk_colsum_splits": (C.c_int, [C.c_int]),
    "ldmk_colsum": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp, _fp]),
    "ldmk_sumpool2": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_resample2": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_silu": (C.c_int, [_fp, _fp, C.c_longlong, _fp]),
    "ldmk_silu_bwd": (C.c_int, [_fp, _fp, _fp, C.c_longlong, _fp]),
    "ldmk_axpy": (C.c_int, [_fp, _fp, C.c_float, C.c_longlong, _fp]),
    "ldmk_q_sample": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp]),
    "ldmk_mse_grad": (C.c_int, [_fp, _fp, _fp, C.c_longlong, C.c_longlong, _fp, _fp, _fp]),
    "ldmk_attn_self_lse": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_attn_self_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_attn_self_lse_bf16": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_attn_self_bwd_bf16": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "ldmk_attn_cross_bwd": (C.c_int, [_fp, C.c_int, _fp, _fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_float, _fp]),
    "ldmk_audio_attention_grad_elems": (C.c_longlong, [C.c_int, C.c_int]),
    "ldmk_audio_attention_bwd": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]),
    "ldmk_head_permute": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ldmk_adamw": (C.c_int, [_fp, _fp, _fp, _fp, C.c_longlong, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                             C.c_int, _fp]),
    "ldmk_ema": (C.c_int, [_fp, _fp, C.c_longlong, C.c_float, _fp]),
}
# every symbol include/ldmk.h declares (checked by tests/test_abi.py against the header text)
EXPORTED = [k for k in _SIGS if k not in ("ldmk_igemm_force_config", "ldmk_attn_force_qt")]

_lib = None


class LdmkError(RuntimeError):
    pass


def load():
    """Load libldmk.so; raises LdmkError when it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LdmkError(f"{LIB_PATH} is missing: run `python -m dsml_thesis_amd.build` (hipcc, gfx950). "
                        "There is no CPU fallback for the sampling path.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64; importing torch first makes the
    # dynamic loader resolve libldmk.so's libamdhip64 dependency to that same, already-loaded runtime, so the
    # streams / device pointers torch hands us are valid in our launches.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if switches.get("LDMK_ATTN_QT"):          # A/B hook: query tiles per wave of the staged attention kernel (1 / 2)
        lib.ldmk_attn_force_qt(int(switches.get("LDMK_ATTN_QT")))
    _lib = lib
    return lib


_inited = set()


def init(device_index):
    """ldmk_init once per device: arch check (gfx950) + kernel attributes set up front, so that no later call -- possibly
    inside a hipGraph capture -- changes a function attribute."""
    device_index = int(device_index)
    if device_index not in _inited:
        check(load().ldmk_init(device_index), "ldmk_init")
        _inited.add(device_index)


def check(rc, what=""):
    if rc != 0:
        msg = load().ldmk_last_error().decode(errors="replace")
        raise LdmkError(f"{what or 'ldmk call'} failed (rc={rc}): {msg}")


def call(name, *args):
    """Call an int-returning entry point and raise on a non-zero code."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        check(rc, name)
