"""ctypes binding of include/elo.h.  There is NO fallback: if libelo_hip.so is
missing or a call fails, the operators raise."""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# ELO_DENSE_F32=1: the comparison build (true fp32 MFMA instead of the fp16 hi/lo split products; build.py)
LIB_PATH = os.environ.get("ELO_LIB_PATH") or os.path.join(      # ELO_LIB_PATH: a debugging build (tools/cv1_clock.sh)
    _PKG, "libelo_hip_f32.so" if os.environ.get("ELO_DENSE_F32", "0") == "1" else "libelo_hip.so")

_vp = ctypes.c_void_p


class GroupArgs(ctypes.Structure):
    """struct elo_group_args (include/elo.h)."""
    _fields_ = [("batch", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
                ("H2", ctypes.c_int), ("W2", ctypes.c_int), ("npoints", ctypes.c_int),
                ("kernel_h", ctypes.c_int), ("kernel_w", ctypes.c_int), ("K", ctypes.c_int),
                ("flag_copy", ctypes.c_int), ("distance", ctypes.c_float),
                ("stride_h", ctypes.c_int), ("stride_w", ctypes.c_int),
                ("xyz1", _vp), ("xyz2", _vp), ("idx_n2", _vp), ("random_hw", _vp),
                ("selected_bhw_idx", _vp), ("valid_idx", _vp), ("valid_in_dis_idx", _vp),
                ("selected_mask", _vp)]


def _struct(name, fields):
    return type(name, (ctypes.Structure,), {"_fields_": fields, "__doc__": "struct %s (include/elo.h)." % name})


_i, _f = ctypes.c_int, ctypes.c_float
GroupConcatArgs = _struct("elo_group_concat_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("H2", _i), ("W2", _i), ("C", _i),
    ("centre_xyz", _vp), ("src_xyz", _vp), ("src_feat", _vp), ("idx", _vp), ("mask", _vp), ("out", _vp)])
MaskedMaxpoolArgs = _struct("elo_masked_maxpool_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("C", _i), ("x", _vp), ("mask", _vp), ("out", _vp)])
CvEncode1Args = _struct("elo_cv_encode1_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("H2", _i), ("W2", _i), ("C", _i),
    ("xyz1", _vp), ("feat1", _vp), ("xyz2", _vp), ("feat2", _vp), ("idx", _vp), ("mask", _vp), ("out", _vp),
    ("dtype", _i)])
CvEncode2Args = _struct("elo_cv_encode2_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("H", _i), ("W", _i), ("C", _i), ("Cc", _i),
    ("xyz1", _vp), ("feat1", _vp), ("cost", _vp), ("idx", _vp), ("mask", _vp), ("xyz_cat", _vp), ("rest", _vp),
    ("dtype", _i)])
SoftmaxPoolArgs = _struct("elo_softmax_pool_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("C", _i), ("logits", _vp), ("values", _vp),
    ("values_stride", _i), ("mask", _vp), ("out", _vp), ("dtype", _i)])
ELO_F32, ELO_F16 = 0, 1
SoftmaxValidArgs = _struct("elo_softmax_valid_args", [
    ("batch", _i), ("npoints", _i), ("C", _i), ("feature", _vp), ("weight", _vp), ("xyz", _vp), ("out", _vp),
    ("scratch", _vp), ("stats", _vp)])
SV_MAX_PARTS = 512     # ELO_SV_MAX_PARTS
PermRefreshArgs = _struct("elo_perm_refresh_args", [
    ("pool", _vp), ("versions", _i), ("total", _i), ("cursor", _vp), ("flat", _vp), ("decoded", _vp), ("entry_of", _vp),
    ("table", _vp), ("n_entries", _i)])
PoseHeadArgs = _struct("elo_pose_head_args", [
    ("batch", _i), ("npoints", _i), ("C", _i), ("hidden", _i), ("feature", _vp), ("weight", _vp), ("xyz", _vp),
    ("W_big", _vp), ("b_big", _vp), ("W_q", _vp), ("b_q", _vp), ("W_t", _vp), ("b_t", _vp),
    ("q_coarse", _vp), ("t_coarse", _vp), ("q", _vp), ("t", _vp), ("q_norm", _vp), ("scratch", _vp), ("pose7", _vp),
    ("clear_scratch", _vp), ("clear_xyz", _vp), ("clear_feat", _vp), ("clear_cells", ctypes.c_long), ("clear_C", _i),
    ("feat_dtype", _i), ("pose7_slots", _i), ("pose7_cursor", _vp), ("next_orders", PermRefreshArgs), ("ready_parts", _i)])
WarpProjectArgs = _struct("elo_warp_project_args", [
    ("batch", _i), ("npoints", _i), ("C", _i), ("H", _i), ("W", _i),
    ("az_res", _f), ("vert_res", _f), ("vert_off", _f),
    ("xyz", _vp), ("feat", _vp), ("q", _vp), ("t", _vp), ("warped", _vp), ("out_xyz", _vp), ("out_feat", _vp),
    ("scratch", _vp), ("prepared", _i), ("feat_dtype", _i)])

InputStageArgs = _struct("elo_input_stage_args", [
    ("batch", _i), ("npoints", _i), ("point_stride", _i), ("H", _i), ("W", _i), ("az_res", _f), ("vert_res", _f),
    ("vert_off", _f), ("crop_xy", _f), ("cloud", _vp), ("T_trans", _vp), ("aug_frame", _vp), ("points", _vp),
    ("out_xyz", _vp), ("scratch", _vp)])

# backward passes (csrc/elo_backward.hip)
GroupConcatBwdArgs = _struct("elo_group_concat_bwd_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("H2", _i), ("W2", _i), ("C", _i),
    ("grad_out", _vp), ("idx", _vp), ("mask", _vp), ("grad_centre", _vp), ("grad_src_xyz", _vp), ("grad_src_feat", _vp)])
MaskedMaxpoolBwdArgs = _struct("elo_masked_maxpool_bwd_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("C", _i), ("x", _vp), ("mask", _vp), ("grad_out", _vp), ("grad_x", _vp)])
CvEncode1BwdArgs = _struct("elo_cv_encode1_bwd_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("H2", _i), ("W2", _i), ("C", _i),
    ("xyz1", _vp), ("xyz2", _vp), ("idx", _vp), ("mask", _vp), ("grad_out", _vp),
    ("grad_xyz1", _vp), ("grad_feat1", _vp), ("grad_xyz2", _vp), ("grad_feat2", _vp)])
CvEncode2BwdArgs = _struct("elo_cv_encode2_bwd_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("H", _i), ("W", _i), ("C", _i), ("Cc", _i),
    ("xyz1", _vp), ("idx", _vp), ("mask", _vp), ("grad_xyz_cat", _vp), ("grad_rest", _vp),
    ("grad_xyz1", _vp), ("grad_feat1", _vp), ("grad_cost", _vp)])
SoftmaxPoolBwdArgs = _struct("elo_softmax_pool_bwd_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("C", _i), ("logits", _vp), ("values", _vp), ("values_stride", _i),
    ("mask", _vp), ("grad_out", _vp), ("grad_logits", _vp), ("grad_values", _vp)])
SoftmaxValidBwdArgs = _struct("elo_softmax_valid_bwd_args", [
    ("batch", _i), ("npoints", _i), ("C", _i), ("feature", _vp), ("weight", _vp), ("xyz", _vp), ("grad_out", _vp),
    ("grad_feature", _vp), ("grad_weight", _vp), ("out", _vp), ("stats", _vp)])
WarpProjectBwdArgs = _struct("elo_warp_project_bwd_args", [
    ("batch", _i), ("npoints", _i), ("C", _i), ("H", _i), ("W", _i), ("az_res", _f), ("vert_res", _f), ("vert_off", _f),
    ("xyz", _vp), ("q", _vp), ("t", _vp), ("scratch", _vp), ("grad_out_xyz", _vp), ("grad_out_feat", _vp),
    ("grad_warped", _vp), ("grad_xyz", _vp), ("grad_feat", _vp), ("grad_q", _vp), ("grad_t", _vp)])

BnStatsArgs = _struct("elo_bn_stats_args", [
    ("rows", ctypes.c_long), ("C", _i), ("z", _vp), ("scratch", _vp), ("eps", _f), ("momentum", _f), ("mean", _vp), ("invstd", _vp),
    ("running_mean", _vp), ("running_var", _vp), ("groups", _i)])
BnApplyArgs = _struct("elo_bn_apply_args", [
    ("rows", ctypes.c_long), ("C", _i), ("z", _vp), ("mean", _vp), ("invstd", _vp), ("gamma", _vp), ("beta", _vp), ("relu", _i),
    ("y", _vp), ("groups", _i)])
BnBackwardArgs = _struct("elo_bn_backward_args", [
    ("rows", ctypes.c_long), ("C", _i), ("dy", _vp), ("z", _vp), ("mean", _vp), ("invstd", _vp), ("gamma", _vp), ("beta", _vp),
    ("relu", _i), ("scratch", _vp), ("sums", _vp), ("dz", _vp), ("groups", _i)])
AdamFlatArgs = _struct("elo_adam_flat_args", [
    ("n", ctypes.c_long), ("param", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("hyper", _vp),
    ("beta1", _f), ("beta2", _f), ("one_minus_beta1", _f), ("one_minus_beta2", _f)])
PoseComposeArgs = _struct("elo_pose_compose_args", [
    ("batch", _i), ("q_raw", _vp), ("t_det", _vp), ("q_coarse", _vp), ("t_coarse", _vp), ("q", _vp), ("t", _vp), ("q_norm", _vp),
    ("grad_q", _vp), ("grad_t", _vp), ("grad_q_norm", _vp), ("grad_q_raw", _vp), ("grad_t_det", _vp), ("grad_q_coarse", _vp),
    ("grad_t_coarse", _vp)])
PoseLossArgs = _struct("elo_pose_loss_args", [
    ("batch", _i), ("q", _vp * 4), ("t", _vp * 4), ("q_gt", _vp), ("t_gt", _vp), ("w_x", _vp), ("w_q", _vp), ("loss", _vp),
    ("grad_out", _vp), ("grad_q", _vp * 4), ("grad_t", _vp * 4), ("grad_w_x", _vp), ("grad_w_q", _vp)])
WeightGradArgs = _struct("elo_weight_grad_args", [
    ("rows", ctypes.c_long), ("Cin", _i), ("Cout", _i), ("x", _vp), ("g", _vp), ("dW", _vp), ("db", _vp), ("scratch", _vp)])
DenseRowsArgs = _struct("elo_dense_rows_args", [
    ("rows", ctypes.c_long), ("Cin", _i), ("Cout", _i), ("x", _vp), ("W", _vp), ("transposed", _i), ("bias", _vp), ("out", _vp),
    ("scratch", _vp), ("eps", _f), ("momentum", _f), ("mean", _vp), ("invstd", _vp), ("running_mean", _vp), ("running_var", _vp),
    ("bn_z", _vp), ("bn_mean", _vp), ("bn_invstd", _vp), ("bn_gamma", _vp), ("bn_beta", _vp), ("bn_sums", _vp), ("bn_relu", _i), ("bn_dz", _vp), ("groups", _i)])
Dense = _struct("elo_dense", [("w_packed", _vp), ("bias", _vp), ("K", _i), ("N", _i), ("relu", _i), ("w_plain", _vp),
                              ("products", _i)])
_l = ctypes.c_long
GroupSpec = _struct("elo_group_spec", [
    ("random_hw", _vp), ("kernel_h", _i), ("kernel_w", _i), ("distance", _f), ("stride_h", _i), ("stride_w", _i),
    ("idx_out", _vp), ("mask_out", _vp), ("decoded_hw", _vp)])
SetconvArgs = _struct("elo_setconv_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("H", _i), ("W", _i), ("H2", _i), ("W2", _i), ("C", _i),
    ("xyz1_grid", _vp), ("centre_hw", _vp), ("centre_xyz", _vp), ("src_xyz", _vp), ("src_feat", _vp),
    ("idx", _vp), ("mask", _vp), ("n_layers", _i), ("layers", Dense * 3), ("out", _vp), ("new_xyz", _vp), ("group", GroupSpec),
    ("feat_dtype", _i), ("range_counter", _vp)])
MlpArgs = _struct("elo_mlp_args", [
    ("rows", _l), ("n_sources", _i), ("src", _vp * 3), ("src_width", _i * 3), ("n_layers", _i),
    ("layers", Dense * 3), ("out", _vp),
    ("n_layers2", _i), ("layers2", Dense * 3), ("before", _vp), ("w_before", _i), ("after", _vp), ("w_after", _i),
    ("out2", _vp), ("feat_dtype", _i),
    ("clear_scratch", _vp), ("clear_xyz", _vp), ("clear_feat", _vp), ("clear_cells", _l), ("clear_C", _i), ("clear_images", _i),
    ("batch_hint", _i), ("sv_scratch", _vp), ("sv_xyz", _vp), ("sv_feature", _vp), ("sv_npoints", _i),
    ("range_counter", _vp)])
Cv1Args = _struct("elo_cv1_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("H2", _i), ("W2", _i), ("C", _i),
    ("xyz1", _vp), ("feat1", _vp), ("xyz2", _vp), ("feat2", _vp), ("idx", _vp), ("mask", _vp),
    ("cv0", Dense), ("cv1", Dense), ("cv2", Dense), ("cv_xyz", Dense), ("sum_cv0", Dense), ("sum_cv1", Dense),
    ("out", _vp), ("group", GroupSpec), ("feat_dtype", _i), ("range_counter", _vp)])
Cv2Args = _struct("elo_cv2_args", [
    ("batch", _i), ("npoints", _i), ("K", _i), ("H", _i), ("W", _i), ("C", _i),
    ("xyz1", _vp), ("feat1", _vp), ("cost", _vp), ("idx", _vp), ("mask", _vp),
    ("xyz_enc", Dense), ("sum_cost0", Dense), ("sum_cost1", Dense), ("out", _vp), ("group", GroupSpec), ("feat_dtype", _i),
    ("range_counter", _vp)])
Tuning = _struct("elo_tuning", [
    ("chain_forms", _i), ("narrow_mfma", _i), ("range_check", _i), ("select_dense_waves", _i), ("random_dense_rows", _i),
    ("setconv_chain_rows", _l), ("mlp_chain_rows", _l), ("small_tile_units", _l), ("pool_wave", _i)])

# every symbol include/elo.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("elo_abi_version", ctypes.c_int, []),
    ("elo_dense_f32", ctypes.c_int, []),
    ("elo_last_error", ctypes.c_char_p, []),
    ("elo_range_check", ctypes.c_int, [ctypes.c_int]),
    ("elo_range_violations", ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), _vp]),
    ("elo_get_tuning", ctypes.c_int, [ctypes.POINTER(Tuning)]),
    ("elo_set_tuning", ctypes.c_int, [ctypes.POINTER(Tuning)]),
    ("elo_fused_conv_random_k", ctypes.c_int, [ctypes.POINTER(GroupArgs), _vp]),
    ("elo_fused_conv_select_k", ctypes.c_int, [ctypes.POINTER(GroupArgs), _vp]),
    ("elo_fused_conv_random_k_dense", ctypes.c_int, [ctypes.POINTER(GroupArgs), _vp]),
    ("elo_fused_conv_select_k_dense", ctypes.c_int, [ctypes.POINTER(GroupArgs), _vp]),
    ("elo_debug_select_dense_waves", ctypes.c_int, [ctypes.c_int]),
    ("elo_perm_refresh", ctypes.c_int, [ctypes.POINTER(PermRefreshArgs), _vp]),
    ("elo_group_concat", ctypes.c_int, [ctypes.POINTER(GroupConcatArgs), _vp]),
    ("elo_masked_maxpool", ctypes.c_int, [ctypes.POINTER(MaskedMaxpoolArgs), _vp]),
    ("elo_cv_encode1", ctypes.c_int, [ctypes.POINTER(CvEncode1Args), _vp]),
    ("elo_cv_encode2", ctypes.c_int, [ctypes.POINTER(CvEncode2Args), _vp]),
    ("elo_masked_softmax_pool", ctypes.c_int, [ctypes.POINTER(SoftmaxPoolArgs), _vp]),
    ("elo_softmax_valid", ctypes.c_int, [ctypes.POINTER(SoftmaxValidArgs), _vp]),
    ("elo_pose_head", ctypes.c_int, [ctypes.POINTER(PoseHeadArgs), _vp]),
    ("elo_warp_project", ctypes.c_int, [ctypes.POINTER(WarpProjectArgs), _vp]),
    ("elo_input_stage", ctypes.c_int, [ctypes.POINTER(InputStageArgs), _vp]),
    ("elo_pose_head_warp", ctypes.c_int, [ctypes.POINTER(PoseHeadArgs), ctypes.POINTER(WarpProjectArgs), _vp]),
    ("elo_group_concat_backward", ctypes.c_int, [ctypes.POINTER(GroupConcatBwdArgs), _vp]),
    ("elo_masked_maxpool_backward", ctypes.c_int, [ctypes.POINTER(MaskedMaxpoolBwdArgs), _vp]),
    ("elo_cv_encode1_backward", ctypes.c_int, [ctypes.POINTER(CvEncode1BwdArgs), _vp]),
    ("elo_cv_encode2_backward", ctypes.c_int, [ctypes.POINTER(CvEncode2BwdArgs), _vp]),
    ("elo_masked_softmax_pool_backward", ctypes.c_int, [ctypes.POINTER(SoftmaxPoolBwdArgs), _vp]),
    ("elo_softmax_valid_backward", ctypes.c_int, [ctypes.POINTER(SoftmaxValidBwdArgs), _vp]),
    ("elo_warp_project_backward", ctypes.c_int, [ctypes.POINTER(WarpProjectBwdArgs), _vp]),
    ("elo_bn_stats", ctypes.c_int, [ctypes.POINTER(BnStatsArgs), _vp]),
    ("elo_bn_apply", ctypes.c_int, [ctypes.POINTER(BnApplyArgs), _vp]),
    ("elo_bn_backward", ctypes.c_int, [ctypes.POINTER(BnBackwardArgs), _vp]),
    ("elo_dense_weight_grad", ctypes.c_int, [ctypes.POINTER(WeightGradArgs), _vp]),
    ("elo_dense_rows", ctypes.c_int, [ctypes.POINTER(DenseRowsArgs), _vp]),
    ("elo_dense_rows_supported", ctypes.c_int, [ctypes.c_long, ctypes.c_int, ctypes.c_int]),
    ("elo_dense_rows_scratch_floats", ctypes.c_long, [ctypes.c_int, ctypes.c_int]),
    ("elo_bn_scratch_floats", ctypes.c_long, [ctypes.c_int, ctypes.c_int]),
    ("elo_adam_flat", ctypes.c_int, [ctypes.POINTER(AdamFlatArgs), _vp]),
    ("elo_pose_compose", ctypes.c_int, [ctypes.POINTER(PoseComposeArgs), _vp]),
    ("elo_pose_loss", ctypes.c_int, [ctypes.POINTER(PoseLossArgs), _vp]),
    ("elo_weight_grad_slices", ctypes.c_int, [ctypes.c_long, ctypes.c_int, ctypes.c_int]),
    ("elo_setconv_fused", ctypes.c_int, [ctypes.POINTER(SetconvArgs), _vp]),
    ("elo_mlp_fused", ctypes.c_int, [ctypes.POINTER(MlpArgs), _vp]),
    ("elo_setconv_fused2", ctypes.c_int, [ctypes.POINTER(SetconvArgs), ctypes.POINTER(SetconvArgs), _vp]),
    ("elo_mlp_fused2", ctypes.c_int, [ctypes.POINTER(MlpArgs), ctypes.POINTER(MlpArgs), _vp]),
    ("elo_mlp_sv_parts", ctypes.c_int, [ctypes.POINTER(MlpArgs), ctypes.POINTER(MlpArgs)]),
    ("elo_cv_stage1_fused", ctypes.c_int, [ctypes.POINTER(Cv1Args), _vp]),
    ("elo_debug_cv1_rr", ctypes.c_int, [ctypes.c_int]),
    ("elo_debug_rr_rows", ctypes.c_int, [ctypes.c_long, ctypes.c_long]),
    ("elo_debug_rr_launches", ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]),
    ("elo_debug_narrow_mfma", ctypes.c_int, [ctypes.c_int]),
    ("elo_debug_narrow_launches", ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]),
    ("elo_debug_sv_ride_launches", ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]),
    ("elo_get_tuning_base", ctypes.c_int, [ctypes.POINTER(Tuning)]),
    ("elo_graph_submit", ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_ulong, _vp, _vp, ctypes.c_int]),
    ("elo_debug_chain_pair_launches", ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]),
    ("elo_cv_stage1_setconv_chain_form", ctypes.c_int, [ctypes.POINTER(Cv1Args), ctypes.POINTER(SetconvArgs), ctypes.POINTER(SetconvArgs)]),
    ("elo_cv_stage1_setconv_chain", ctypes.c_int, [ctypes.POINTER(Cv1Args), ctypes.POINTER(SetconvArgs), ctypes.POINTER(SetconvArgs), _vp]),
    ("elo_cv_stage2_fused", ctypes.c_int, [ctypes.POINTER(Cv2Args), _vp]),
    ("elo_cv_stage1_setconv_fused", ctypes.c_int, [ctypes.POINTER(Cv1Args), ctypes.POINTER(SetconvArgs), ctypes.POINTER(SetconvArgs), _vp]),
]

_lib = None


class EloError(RuntimeError):
    pass


ABI_VERSION = 26


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EloError("libelo_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` -- there is no CPU fallback for the hot path" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(handle, name)          # AttributeError if the symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.elo_abi_version() != ABI_VERSION:      # the ctypes structs below mirror ONE layout of include/elo.h
            raise EloError("%s has ABI version %d, this host expects %d: rebuild (python -c 'import __graft_entry__ as g; "
                           "g.build()')" % (LIB_PATH, handle.elo_abi_version(), ABI_VERSION))
        _lib = handle
        from . import tuning
        env = tuning.lib_from_env()                      # the ONE place the library's forms meet the environment
        if env:
            set_tuning(**env)
    return _lib


def get_tuning():
    """The library's elo_tuning as a dict (include/elo.h)."""
    t = Tuning()
    check(lib().elo_get_tuning(ctypes.byref(t)))
    return {name: int(getattr(t, name)) for name, _ in Tuning._fields_}


def get_tuning_base():
    """What elo_set_tuning last installed (without pending elo_debug_* overrides)."""
    t = Tuning()
    check(lib().elo_get_tuning_base(ctypes.byref(t)))
    return {name: int(getattr(t, name)) for name, _ in Tuning._fields_}


def set_tuning(**fields):
    """Change fields of the library's elo_tuning (the others keep their value); raises on a value outside its domain.
    The read-modify-write starts from what elo_set_tuning last installed (elo_get_tuning_base), not from the launchers' view: a pending
    elo_debug_* override is not promoted into the base by an unrelated change."""
    t = Tuning()
    check(lib().elo_get_tuning_base(ctypes.byref(t)))
    for name, value in fields.items():
        if not hasattr(t, name):
            raise KeyError("elo_tuning has no field %r" % name)
        setattr(t, name, int(value))
    check(lib().elo_set_tuning(ctypes.byref(t)))
    from . import tuning
    tuning.bump()


def check(rc):
    if rc != 0:
        raise EloError("libelo_hip: %s (status %d)" % (lib().elo_last_error().decode(), rc))


# The device word the CHECKED fused kernels of the launches below add their out-of-range operands to (None: the process-wide
# counter).  model.capture sets it to the lane's own word while it records that lane's checked graph: the pointer is baked into
# the graph's kernel arguments, so a violation is the lane's that ran the forward, not every lane's.
_range_counter = None


def set_range_counter(ptr):
    """ptr: device address (int) of an unsigned 64-bit word, or None; returns the previous value."""
    global _range_counter
    prev, _range_counter = _range_counter, ptr
    return prev


def _stamp(*structs):
    if _range_counter is not None:
        for a in structs:
            if a is not None and hasattr(a, "range_counter"):
                a.range_counter = _range_counter


def call(entry, args, like):
    """Launch `entry(args, current stream)` on like.device; raise on a non-zero status."""
    _stamp(args)
    with torch.cuda.device(like.device):
        check(getattr(lib(), entry)(ctypes.byref(args), stream_ptr(like)))


def call3(entry, a, b, c, like):
    """`entry(a, b, c or NULL, current stream)`."""
    _stamp(a, b, c)
    with torch.cuda.device(like.device):
        check(getattr(lib(), entry)(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c) if c is not None else None, stream_ptr(like)))


def call2(entry, args_a, args_b, like):
    """Paired launch: `entry(a, b, current stream)`."""
    _stamp(args_a, args_b)
    with torch.cuda.device(like.device):
        check(getattr(lib(), entry)(ctypes.byref(args_a), ctypes.byref(args_b), stream_ptr(like)))


def call2n(entry, args_a, args_b, like):
    """Paired launch whose second job is optional: `entry(a, b or NULL, current stream)`."""
    _stamp(args_a, args_b)
    with torch.cuda.device(like.device):
        check(getattr(lib(), entry)(ctypes.byref(args_a), ctypes.byref(args_b) if args_b is not None else None, stream_ptr(like)))


def stream_ptr(t):
    """hipStream_t of torch's current stream on t's device, as void*."""
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise EloError("the EfficientLO-Net hot path runs on an AMD GPU only (got a %s tensor); "
                           "there is no CPU fallback" % t.device)


def dtype_code(t):
    """ELO_F32 / ELO_F16 of a feature tensor (include/elo.h feat_dtype)."""
    if t.dtype == torch.float32:
        return ELO_F32
    if t.dtype == torch.float16:
        return ELO_F16
    raise TypeError("feature tensors are float32 or float16 (got %s)" % t.dtype)


def range_check(enable):
    """Switch the operand range check of the fused kernels (include/elo.h elo_range_check); returns the previous setting."""
    return lib().elo_range_check(1 if enable else 0)


def range_violations(like):
    """Operands with |x| >= 65504 or NaN seen by the checked kernels since the last call (synchronises the stream)."""
    n = ctypes.c_ulonglong(0)
    with torch.cuda.device(like.device):
        check(lib().elo_range_violations(ctypes.byref(n), stream_ptr(like)))
    return int(n.value)
