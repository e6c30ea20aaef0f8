"""Loader + ctypes prototypes for libganslate_hip.so.

The library is the product path: if it is missing or fails to load, importing the ops raises — there is no
CPU fallback (the CPU oracle under oracle/ is test infrastructure only)."""
import ctypes as C
import os
from pathlib import Path

GS_MAX_TAPS = 352
BORDER = {"zero": 0, "reflect": 1, "replicate": 2}
ACT = {"none": 0, "relu": 1, "lrelu": 2, "tanh": 3}


class GConvDesc(C.Structure):
    """Mirror of gs_gconv_desc (include/ganslate_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "N", "Hi", "Wi", "Ci", "Di", "Do", "Dc", "pz", "in_cs", "in_co", "Ho", "Wo", "Co", "out_cs", "out_co", "Hc", "Wc",
        "so", "py", "px", "si", "T", "Kp", "w_rows", "border", "act")] + [
        ("slope", C.c_float), ("stats_slots", C.c_int32), ("stats_slot0", C.c_int32), ("accumulate", C.c_int32),
        ("dh", C.c_int8 * GS_MAX_TAPS), ("dw", C.c_int8 * GS_MAX_TAPS), ("dd", C.c_int8 * GS_MAX_TAPS)]


class WGradDesc(C.Structure):
    """Mirror of gs_wgrad_desc."""
    _fields_ = [(n, C.c_int32) for n in (
        "N", "Ha", "Wa", "P", "a_cs", "a_co", "Hg", "Wg", "Q", "g_cs", "g_co", "Da", "Dg", "si", "T", "border",
        "dw_ld", "dw_fresh")] + [
        ("dh", C.c_int8 * GS_MAX_TAPS), ("dw_", C.c_int8 * GS_MAX_TAPS), ("dd", C.c_int8 * GS_MAX_TAPS)]


class NormExDesc(C.Structure):
    """Mirror of gs_norm_ex_desc."""
    _fields_ = [(n, C.c_int32) for n in ("N", "H", "W", "C", "act1", "act2")] + [("slope", C.c_float)] + [
        (n, C.c_int32) for n in ("x1_cs", "x1_co", "x2_cs", "x2_co", "g1_cs", "g1_co", "g2_cs", "g2_co")] + [
        ("drop_p", C.c_float), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32), ("seed_dev", C.c_void_p)]


class GConvFuse(C.Structure):
    """Mirror of gs_gconv_fuse."""
    _fields_ = [("y", C.c_void_p), ("mean_rstd", C.c_void_p), ("g2", C.c_void_p), ("partial", C.c_void_p)] + [
        (n, C.c_int32) for n in ("Dy", "Hy", "Wy", "fold", "fold_mode", "act")] + [("slope", C.c_float)]


class Twin(C.Structure):
    """Mirror of gs_twin."""
    _fields_ = [("n_split", C.c_int32), ("pad_", C.c_int32), ("w_delta", C.c_int64), ("bias_delta", C.c_int64),
                ("dw_delta", C.c_int64)]


class AdamFuse(C.Structure):
    """Mirror of gs_adam_fuse."""
    _fields_ = [("p", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("hyper", C.c_void_p), ("inv_f", C.c_void_p),
                ("fpack", C.c_void_p), ("inv_d", C.c_void_p), ("dpack", C.c_void_p), ("tr_base", C.c_void_p),
                ("tr_kp", C.c_void_p), ("tr_pack", C.c_void_p)]


class NormDbItem(C.Structure):
    """Mirror of gs_norm_db_item."""
    _fields_ = [("sums", C.c_void_p), ("mean_rstd", C.c_void_p), ("db", C.c_void_p), ("N", C.c_int32), ("C", C.c_int32),
                ("inv_hw", C.c_float), ("pad_", C.c_int32)]


class PatchNCEDesc(C.Structure):
    """Mirror of gs_patchnce_desc."""
    _fields_ = [("levels", C.c_int32), ("batch", C.c_int32), ("patches", C.c_int32), ("nc", C.c_int32),
                ("channels", C.c_int32 * 8), ("nce_T", C.c_float), ("lambda_nce", C.c_float)]


class AttnDesc(C.Structure):
    """Mirror of gs_attn_desc."""
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("C", C.c_int32)]


class AttnParams(C.Structure):
    """Mirror of gs_attn_params (device pointers to fp32 tensors in torch layout)."""
    _fields_ = [(n, C.c_void_p) for n in ("gamma", "wq", "bq", "wk", "bk", "wv", "bv")]


class PNormDesc(C.Structure):
    """Mirror of gs_pnorm_desc."""
    _fields_ = [("pixels", C.c_int64)] + [(n, C.c_int32) for n in (
        "N", "C", "y_cs", "y_co", "res_mode", "res_cs", "res_co", "res_mod", "out_cs", "out_co", "g_cs", "g_co",
        "g2_cs", "g2_co", "dy_cs", "dy_co", "gres_cs", "gres_co")]


_PROTOS = {
    "gs_init": (C.c_int, [C.c_int]),
    "gs_shutdown": (None, []),
    "gs_last_error": (C.c_char_p, []),
    "gs_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "gs_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "gs_tile_m": (C.c_int, [C.POINTER(GConvDesc)]),
    "gs_gconv_stat_slots": (C.c_int, [C.POINTER(GConvDesc)]),
    "gs_gconv_ring_slots": (C.c_int, [C.POINTER(GConvDesc)]),
    "gs_gconv_ring_apply_words": (C.c_int, [C.POINTER(GConvDesc)]),
    "gs_gconv_ring_apply": (C.c_int, [C.POINTER(GConvDesc), C.c_void_p, C.c_void_p, C.POINTER(GConvFuse), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.POINTER(Twin), C.c_void_p]),
    "gs_gconv_forward": (C.c_int, [C.POINTER(GConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "gs_gconv_splitk_ws_floats": (C.c_int64, [C.POINTER(GConvDesc)]),
    "gs_gconv_forward_ws": (C.c_int, [C.POINTER(GConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int64, C.c_void_p]),
    "gs_gconv_forward_multi": (C.c_int, [C.POINTER(C.POINTER(GConvDesc)), C.c_int32, C.c_void_p, C.POINTER(C.c_void_p),
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gs_gconv_multi_splitk_ws_floats": (C.c_int64, [C.POINTER(C.POINTER(GConvDesc)), C.c_int32]),
    "gs_gconv_forward_multi_ws": (C.c_int, [C.POINTER(C.POINTER(GConvDesc)), C.c_int32, C.c_void_p, C.POINTER(C.c_void_p),
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gs_gconv_multi_fused_slots": (C.c_int, [C.c_void_p, C.c_int32]),
    "gs_gconv_forward_multi_fused": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.POINTER(GConvFuse), C.c_void_p]),
    "gs_gconv_forward_fused": (C.c_int, [C.POINTER(GConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.POINTER(GConvFuse), C.c_void_p]),
    "gs_gconv_twin_native": (C.c_int, [C.POINTER(GConvDesc), C.c_void_p]),
    "gs_gconv_forward_twin": (C.c_int, [C.POINTER(GConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.POINTER(Twin), C.c_void_p]),
    "gs_gconv_multi_twin_native": (C.c_int, [C.c_void_p, C.c_int32]),
    "gs_gconv_forward_multi_twin": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "gs_wgrad_pair": (C.c_int, [C.POINTER(WGradDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p]),
    "gs_wgrad": (C.c_int, [C.POINTER(WGradDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gs_wgrad_adam_eligible": (C.c_int, [C.POINTER(WGradDesc)]),
    "gs_wgrad_adam": (C.c_int, [C.POINTER(WGradDesc), C.c_void_p, C.c_void_p, C.POINTER(AdamFuse), C.c_void_p]),
    "gs_wgrad_ws_floats": (C.c_int64, [C.POINTER(WGradDesc), C.c_int32]),
    "gs_wgrad_ws": (C.c_int, [C.POINTER(WGradDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_int64, C.c_void_p]),
    "gs_wgrad_twin_native": (C.c_int, [C.POINTER(WGradDesc), C.c_int32]),
    "gs_wgrad_ws_floats_twin": (C.c_int64, [C.POINTER(WGradDesc), C.c_int32]),
    "gs_wgrad_ws_twin": (C.c_int, [C.POINTER(WGradDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.POINTER(Twin), C.c_void_p]),
    "gs_bias_grad": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gs_bias_grad_ws_floats": (C.c_int64, [C.c_int64, C.c_int32]),
    "gs_bias_grad_ws": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_int64, C.c_void_p]),
    "gs_inorm_finalize": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_float,
                                    C.c_void_p, C.c_void_p]),
    "gs_inorm_act_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                       C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "gs_inorm_stats_act_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_float,
                                             C.c_void_p]),
    "gs_inorm_act_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "gs_norm_bias_grads": (C.c_int, [C.POINTER(NormDbItem), C.c_int32, C.c_void_p]),
    "gs_inorm_backward_scratch_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "gs_norm_act_forward_ex": (C.c_int, [C.POINTER(NormExDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "gs_norm_act_backward_ex": (C.c_int, [C.POINTER(NormExDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gs_norm_backward_ex_scratch_floats": (C.c_int64, [C.POINTER(NormExDesc)]),
    "gs_pnorm_forward": (C.c_int, [C.POINTER(PNormDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "gs_pnorm_backward": (C.c_int, [C.POINTER(PNormDesc)] + [C.c_void_p] * 12),
    "gs_pnorm_backward_scratch_floats": (C.c_int64, [C.POINTER(PNormDesc)]),
    "gs_add_views": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                               C.c_int32, C.c_void_p]),
    "gs_repeat_backward": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int64, C.c_void_p]),
    "gs_image_to_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_void_p]),
    "gs_image_pair_to_act": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_void_p]),
    "gs_image_pair_to_act_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                                C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gs_act_to_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_void_p]),
    "gs_act_to_image_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gs_image_to_act_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gs_image_unfold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gs_image_unfold_backward": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 11 + [C.c_void_p]),
    "gs_shiftadd_to_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gs_shiftadd_to_image_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64,
                                                C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gs_slice_stats_slots": (C.c_int32, [C.c_int64]),
    "gs_slice_stats": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gs_u8_resample_h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_int32, C.c_void_p]),
    "gs_u8_resample_v": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_int32, C.c_void_p]),
    "gs_u8_resample_v_crop_normalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                  C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                  C.c_int32, C.c_int32, C.c_void_p]),
    "gs_patch_zscore_ws_floats": (C.c_int64, []),
    "gs_patch_zscore": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "gs_mse_const": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gs_adv_loss": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p]),
    "gs_l1": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gs_mean": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gs_scalar_affine": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32,
                                   C.c_void_p, C.c_void_p]),
    "gs_sum2_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gs_conv_cout1_eligible": (C.c_int, [C.c_void_p]),
    "gs_conv_cout1_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gs_wgrad_cout1_eligible": (C.c_int, [C.c_void_p]),
    "gs_wgrad_cout1_ws_floats": (C.c_int64, [C.c_void_p]),
    "gs_wgrad_cout1_ws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_void_p]),
    "gs_tap_gather": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                C.c_void_p]),
    "gs_tap_scatter_add": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gs_tap_rows_sum": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "gs_zero_bytes": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "gs_flip_w_if": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "gs_image_tap_gather": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_void_p]),
    "gs_image_tap_scatter": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p]),
    "gs_bias_grad_head_ws": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_void_p]),
    "gs_ssim_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "gs_ssim_scratch_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "gs_ssim_distance_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "gs_ssim_backward_scratch_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "gs_patchnce_param_floats": (C.c_int64, [C.POINTER(PatchNCEDesc)]),
    "gs_patchnce_work_bytes": (C.c_int64, [C.POINTER(PatchNCEDesc)]),
    "gs_patchnce_forward": (C.c_int, [C.POINTER(PatchNCEDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "gs_patchnce_backward": (C.c_int, [C.POINTER(PatchNCEDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gs_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_float),
                               C.c_float, C.c_int32, C.c_void_p]),
    "gs_adam_step_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_float, C.c_int32, C.c_void_p]),
    "gs_adam_step_dev_packs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_float,
                                         C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gs_adam_step_dev_packs_ranges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                                C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p]),
    "gs_pool_query": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
    "gs_repack_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gs_repack_bf16_tiled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gs_repack_bf16_groups": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gs_repack_bf16_tiled_groups": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
    "gs_attn_work_bytes": (C.c_int64, [C.POINTER(AttnDesc)]),
    "gs_attn_forward_work_bytes": (C.c_int64, [C.POINTER(AttnDesc)]),
    "gs_attn_forward": (C.c_int, [C.POINTER(AttnDesc), C.c_void_p, C.POINTER(AttnParams), C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "gs_attn_backward": (C.c_int, [C.POINTER(AttnDesc), C.c_void_p, C.c_void_p, C.POINTER(AttnParams),
                                   C.POINTER(AttnParams), C.c_void_p, C.c_void_p, C.c_void_p]),
}

EXPORTS = tuple(_PROTOS)
_lib = None


def library_path() -> Path:
    env = os.environ.get("GANSLATE_HIP_LIB")
    return Path(env) if env else Path(__file__).resolve().parent.parent / "libganslate_hip.so"


def load():
    """dlopen the library and attach prototypes (no GPU call is made here)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.is_file():
        raise RuntimeError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(make -C ganslate_amd/csrc). There is no CPU fallback for the training step.")
    lib = C.CDLL(str(path))
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # raises AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class HipError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        raise HipError(f"{what} failed (rc={rc}): {load().gs_last_error().decode()}")
