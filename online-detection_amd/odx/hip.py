"""ctypes binding of libodx.so (C ABI: include/odx.h).

The product path has no CPU fallback: if the shared library is missing, or no MI355X is
visible, every entry point raises ``OdxUnavailable`` loudly.  Device memory, streams and
collectives come from PyTorch-ROCm; the arithmetic is the hand-written gfx950 HIP in
``online-detection_amd/csrc``.
"""
import ctypes
import os

_LIB = None
# (ODX_LIB_PATH: a development knob — another build of the library)
_LIB_PATH = os.environ.get("ODX_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libodx.so")

ODX_OK = 0
KNM_F32, KNM_U24, KNM_BF16 = 0, 1, 2
GEMM_LOWER_ONLY, GEMM_A_UPPER, GEMM_B_UPPER, GEMM_A_LOWER, GEMM_B_LOWER, GEMM_STORE_T = 1, 2, 4, 8, 16, 32


class OdxUnavailable(RuntimeError):
    pass


class OdxError(RuntimeError):
    pass


_vp, _i64, _i32, _f64, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_float

# name -> (restype, argtypes): exactly the declarations of include/odx.h
SIGNATURES = {
    "odx_last_error_string": (ctypes.c_char_p, []),
    "odx_knm_pass_kernel_name": (ctypes.c_char_p, [_i64, _i32, _i32]),
    "odx_version": (_i32, []),
    "odx_device_cus": (_i32, []),
    "odx_set_option": (_i32, [ctypes.c_char_p, _i32]),
    "odx_get_option": (_i32, [ctypes.c_char_p, _vp]),
    "odx_option_default": (_i32, [ctypes.c_char_p, _vp]),
    "odx_release_helper_streams": (_i32, []),
    "odx_set_helper_streams": (_i32, [_vp, _vp]),
    "odx_stream_create_cu_mask": (_i32, [_vp, _i32, _vp]),
    "odx_stream_destroy": (_i32, [_vp]),
    "odx_set_pass_cus": (_i32, [_i32]),
    "odx_set_side_stream_cu_mask": (_i32, [_vp, _i32]),
    "odx_debug_placement": (_i32, [_vp, _i32, _i32, _vp]),
    "odx_row_sqnorm_f32": (_i32, [_vp, _i64, _i64, _i32, _vp, _vp]),
    "odx_row_sqnorm_absmax_f32": (_i32, [_vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "odx_split_f16_premax": (_i32, [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp]),
    "odx_gauss_knm_f32": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _f64, _vp, _i64, _vp]),
    "odx_gauss_knm_direct_f32": (_i32, [_vp, _i64, _i64, _vp, _i64, _i64, _i32, _f64, _vp, _i64, _vp]),
    "odx_gauss_mmv_f32": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _f64, _vp, _i64, _vp, _i32, _vp, _i64, _vp]),
    "odx_split_f16": (_i32, [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp]),
    "odx_gauss_h2_tile": (_i32, [_i64, _i64]),
    "odx_set_h2_tile": (_i32, [_i32]),
    "odx_gauss_knm_h2": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _f64, _vp, _i64, _vp]),
    "odx_gauss_knm_h2_rhs_workspace_bytes": (_i64, [_i64, _i64]),
    "odx_gauss_knm_h2_rhs": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _f64, _vp, _i64, _vp, _vp, _vp, _i64,
                                    _vp]),
    "odx_gauss_mmv_h2_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "odx_gauss_mmv_h2": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _f64, _vp, _i64, _vp, _i32, _vp, _i64,
                                _vp, _i64, _vp]),
    "odx_falkon_cg_workspace_bytes": (_i64, [_i64, _i64]),
    "odx_falkon_cg_f64": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _f64, _f64, _i32, _i32, _f64, _f64, _vp, _vp,
                                 _i64, _vp]),
    "odx_falkon_cg_batched_workspace_bytes": (_i64, [_i32, _vp, _vp]),
    "odx_falkon_cg_batched_f64": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _f64, _i32, _i32, _f64, _f64,
                                         _vp, _vp, _i64, _vp]),
    "odx_falkon_cg_batched_q_workspace_bytes": (_i64, [_i32, _vp, _vp, _i32]),
    "odx_falkon_cg_batched_q_f64": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _f64, _i32, _i32,
                                           _f64, _f64, _vp, _vp, _i64, _vp]),
    "odx_knm_fwd_bwd_workspace_bytes": (_i64, [_i64, _i64]),
    "odx_set_pass_reserved_cus": (_i32, [_i32]),
    "odx_knm_fwd_bwd": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "odx_knm_fwd_bwd2_workspace_bytes": (_i64, [_i64, _i64]),
    "odx_knm_fwd_bwd2": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "odx_knm_ld": (_i64, [_i64, _i32]),
    "odx_knm_bytes": (_i64, [_i64, _i64, _i32]),
    "odx_gauss_knm_h2_store": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _f64, _i32, _vp, _i64, _vp, _i64,
                                      _vp, _vp, _vp, _i64, _vp]),
    "odx_knm_fwd_bwd_q_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "odx_knm_fwd_bwd_q": (_i32, [_vp, _i64, _vp, _i64, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "odx_knm_fwd_bwd2_q_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "odx_knm_fwd_bwd2_q": (_i32, [_vp, _i64, _vp, _i64, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "odx_split_f8": (_i32, [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _vp]),
    "odx_gauss_knm_f8_store": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _f64, _i32, _vp, _i64, _vp, _i64,
                                      _vp, _vp, _vp, _i64, _vp]),
    "odx_gauss_mmv_f8": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _f64, _vp, _i64, _vp, _i32, _vp, _i64,
                                _vp, _i64, _vp]),
    "odx_cg_residual": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "odx_falkon_precond_workspace_bytes": (_i64, [_i64, _i32]),
    "odx_falkon_precond_f64": (_i32, [_vp, _i64, _i64, _i32, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    "odx_falkon_precond_batched_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "odx_falkon_precond_batched_f64": (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _f64, _f64, _f64, _vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    "odx_trmv_f64": (_i32, [_vp, _i64, _i64, _i32, _vp, _f64, _f64, _vp, _vp, _vp]),
    "odx_cg_init": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "odx_cg_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _f64, _i32, _i64, _vp]),
    "odx_cg_finish": (_i32, [_vp, _vp, _vp, _f64, _f64, _i64, _vp]),
    "odx_axpby_f64": (_i32, [_f64, _vp, _f64, _vp, _i64, _vp]),
    "odx_gemm_nt_f64": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _f64, _f64, _i32, _vp]),
    "odx_gemm_nt_f32": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _f32, _f32, _i32, _vp]),
    "odx_potrf_workspace_bytes": (_i64, [_i64]),
    "odx_potrf_f64": (_i32, [_vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    "odx_trtri_workspace_bytes": (_i64, [_i64]),
    "odx_trtri_f64": (_i32, [_vp, _i64, _i64, _vp, _vp, _i64, _vp, _i64, _vp]),
    "odx_convert_f32_f64": (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    "odx_convert_f64_f32": (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    "odx_rls_gram_workspace_bytes": (_i64, [_i64, _i32]),
    "odx_rls_gram_f64": (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "odx_rls_solve_workspace_bytes": (_i64, [_i32]),
    "odx_rls_solve_f64": (_i32, [_vp, _i64, _i32, _f64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "odx_rls_gram_batched_workspace_bytes": (_i64, [_i64, _i32]),
    "odx_rls_gram_batched_f64": (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "odx_rls_pad_index": (_i32, [_vp, _i64, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "odx_rls_gram_raw_batched_f64": (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "odx_rls_fold_whitened_f64": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp]),
    "odx_rls_xty_batched_f64": (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "odx_rls_rows_form": (_i32, [_vp, _i64, _i32]),
    "odx_rls_solve_batched_workspace_bytes": (_i64, [_i32, _i32]),
    "odx_rls_solve_batched_f64": (_i32, [_vp, _i64, _i64, _i32, _i32, _f64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    "odx_rls_predict_rows_f64": (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "odx_rls_predict_rows_batched_f64": (_i32, [_vp, _i64, _i32, _vp, _vp, _i32, _i64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "odx_roi_align_fwd_f32": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _f32, _i32, _i32, _i32, _vp, _vp]),
    "odx_split_f16_taps3x3": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _i64, _vp, _vp]),
    "odx_gemm_b16": (_i32, [_vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _vp]),
    "odx_taps3x3_16": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "odx_gemm_h2_f32": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    "odx_roi_align_rows_f32": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _f32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "odx_roi_align_rows_nhwc_f32": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _i32, _f32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "odx_gemm_h2_max_f32": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _vp]),
    "odx_split_f16_taps3x3_premax": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _i64, _vp, _vp]),
    "odx_gemm_h2_taps_supported": (_i32, [_i64, _i64, _i32, _i64]),
    "odx_gemm_h2_taps_f32": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _vp, _i64, _vp,
                                    _vp, _i64, _f32, _f32, _vp, _vp]),
    "odx_gemm_h2_chain_f32": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp,
                                     _vp, _i64, _f32, _f32, _vp, _vp]),
    "odx_taps3x3_packed": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "odx_rpn_label_f32": (_i32, [_vp, _i32, _vp, _vp, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "odx_det_label_f32": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _f32, _f32, _f32, _f32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "odx_box_targets_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "odx_debug_gemm_h2_mf32": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp]),
    "odx_gemm_b16_taps_supported": (_i32, [_i64, _i64, _i32, _i64]),
    "odx_gemm_b16_taps": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _vp]),
    "odx_roi_align_fpn_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "odx_roi_align_fpn_nhwc_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "odx_stem_pool_rows_f32": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp]),
    "odx_stem_pool_rows_16": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "odx_upsample_add_rows_f32": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "odx_upsample_add_rows_16": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "odx_roi_align_fpn_nhwc_16": (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "odx_nms_workspace_bytes": (_i64, [_i32]),
    "odx_nms_f32": (_i32, [_vp, _i32, _f32, _vp, _vp, _i64, _vp]),
    "odx_nms_first_f32": (_i32, [_vp, _i32, _f32, _i32, _vp, _vp, _i64, _vp]),
    "odx_nms_batched_workspace_bytes": (_i64, [_i32, _i32]),
    "odx_nms_batched_f32": (_i32, [_vp, _vp, _i32, _i32, _f32, _vp, _vp, _i64, _vp]),
    "odx_rpn_topk_decode_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "odx_nms_compact_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "odx_nms_batched_first_f32": (_i32, [_vp, _vp, _i32, _i32, _f32, _i32, _vp, _vp, _i64, _vp]),
    "odx_paste_masks_u8": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp]),
    "odx_bias_act_nchw_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _i32, _vp]),
    "odx_bias_act_nchw_16": (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _i64, _i32, _vp]),
}


def lib_path():
    return _LIB_PATH


def load():
    """dlopen libodx.so and bind every symbol of include/odx.h (no GPU needed for this)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise OdxUnavailable(
            "libodx.so not found at %s — build it with `make -C online-detection_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback." % _LIB_PATH)
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc, what=""):
    if rc != ODX_OK:
        msg = load().odx_last_error_string().decode("utf-8", "replace")
        raise OdxError("%s failed (code %d): %s" % (what or "libodx call", rc, msg))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise OdxUnavailable("no HIP device visible: the odx hot path runs on MI355X only (no CPU fallback)")
    return load()
