"""ctypes loader for libluminoth_hip.so (the C ABI in include/luminoth_hip.h).

cffi (named by BASELINE.json north_star) is not installed in this image; ctypes
plays the same thin-FFI role.  There is NO fallback: if the library is missing
or a call fails, the product path raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libluminoth_hip.so')
_lib = None

c_f = ctypes.c_void_p   # device pointers travel as void*
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_fl = ctypes.c_float
c_sz = ctypes.c_size_t


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        'N', 'H', 'W', 'C', 'K', 'R', 'S', 'OH', 'OW', 'stride', 'dilation',
        'pad_top', 'pad_left', 'act', 'compute')]


class RpnProposalDesc(ctypes.Structure):
    _fields_ = [('B', ctypes.c_int32), ('feat_h', ctypes.c_int32), ('feat_w', ctypes.c_int32),
                ('A', ctypes.c_int32), ('anchor_stride', ctypes.c_int32),
                ('im_h', ctypes.c_float), ('im_w', ctypes.c_float),
                ('pre_nms_top_n', ctypes.c_int32), ('post_nms_top_n', ctypes.c_int32),
                ('nms_threshold', ctypes.c_float), ('min_prob_threshold', ctypes.c_float),
                ('apply_nms', ctypes.c_int32), ('clip_after_nms', ctypes.c_int32),
                ('filter_outside_anchors', ctypes.c_int32)]


class RpnTargetDesc(ctypes.Structure):
    _fields_ = [('B', ctypes.c_int32), ('feat_h', ctypes.c_int32), ('feat_w', ctypes.c_int32),
                ('A', ctypes.c_int32), ('anchor_stride', ctypes.c_int32), ('Gmax', ctypes.c_int32),
                ('im_h', ctypes.c_int32), ('im_w', ctypes.c_int32),
                ('allowed_border', ctypes.c_int32), ('clobber_positives', ctypes.c_int32),
                ('foreground_threshold', ctypes.c_float), ('background_threshold_high', ctypes.c_float),
                ('foreground_fraction', ctypes.c_float), ('minibatch_size', ctypes.c_int32)]


class RcnnTargetDesc(ctypes.Structure):
    _fields_ = [('B', ctypes.c_int32), ('P', ctypes.c_int32), ('Gmax', ctypes.c_int32),
                ('minibatch_size', ctypes.c_int32),
                ('foreground_fraction', ctypes.c_float), ('foreground_threshold', ctypes.c_float),
                ('background_threshold_high', ctypes.c_float), ('background_threshold_low', ctypes.c_float),
                ('variance_xy', ctypes.c_float), ('variance_wh', ctypes.c_float)]


class RcnnProposalDesc(ctypes.Structure):
    _fields_ = [('B', ctypes.c_int32), ('R', ctypes.c_int32), ('C', ctypes.c_int32),
                ('im_h', ctypes.c_float), ('im_w', ctypes.c_float),
                ('variance_xy', ctypes.c_float), ('variance_wh', ctypes.c_float),
                ('class_max_detections', ctypes.c_int32), ('class_nms_threshold', ctypes.c_float),
                ('total_max_detections', ctypes.c_int32), ('min_prob_threshold', ctypes.c_float),
                ('class_agnostic_boxes', ctypes.c_int32)]


class SsdTargetDesc(ctypes.Structure):
    _fields_ = [('B', ctypes.c_int32), ('N', ctypes.c_int32), ('C', ctypes.c_int32), ('Gmax', ctypes.c_int32),
                ('foreground_threshold', ctypes.c_float), ('background_threshold_high', ctypes.c_float),
                ('hard_negative_ratio', ctypes.c_float), ('variance_xy', ctypes.c_float),
                ('variance_wh', ctypes.c_float)]


class WinoWeightJob(ctypes.Structure):
    _fields_ = [('w', ctypes.c_void_p), ('kscale', ctypes.c_void_p), ('u', ctypes.c_void_p),
                ('C', ctypes.c_int32), ('K', ctypes.c_int32)]


class X3WeightJob(ctypes.Structure):
    _fields_ = [('w', ctypes.c_void_p), ('out', ctypes.c_void_p), ('rs', ctypes.c_int32), ('C', ctypes.c_int32),
                ('K', ctypes.c_int32)]


class HalfWeightJob(ctypes.Structure):
    _fields_ = [('w', ctypes.c_void_p), ('kscale', ctypes.c_void_p), ('w_fwd', ctypes.c_void_p), ('w_bwd', ctypes.c_void_p),
                ('RS', ctypes.c_int32), ('C', ctypes.c_int32), ('K', ctypes.c_int32)]


class WgradTail(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ('slabs', 'dw', 'w', 'scale', 'mean', 'rstd', 'dgamma', 'colpart',
                                                 'colsum')] + \
               [('n', ctypes.c_int64), ('splits', ctypes.c_int32), ('K', ctypes.c_int32), ('colrows', ctypes.c_int32),
                ('reserved', ctypes.c_int32)]


P = ctypes.POINTER
SIGNATURES = {
    # name: (restype, argtypes)   -- lists EVERY symbol include/luminoth_hip.h declares
    'lmh_version': (c_i, []),
    'lmh_last_error': (ctypes.c_char_p, []),
    'lmh_device_count': (c_i, []),
    'lmh_set_option': (c_i, [ctypes.c_char_p, c_i]),
    'lmh_set_default_option': (c_i, [ctypes.c_char_p, c_i]),
    'lmh_get_option': (c_i, [ctypes.c_char_p, P(c_i)]),
    'lmh_conv2d_fwd': (c_i, [P(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    'lmh_conv2d_bwd_data': (c_i, [P(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    'lmh_act_bits': (c_i, [c_f, c_i, c_i64, c_i, c_f, c_f]),
    'lmh_apply_act_bits': (c_i, [c_f, c_f, c_i64, c_i, c_f]),
    'lmh_conv2d_winograd_ok': (c_i, [P(ConvDesc)]),
    'lmh_conv2d_winograd_workspace_bytes': (c_sz, [P(ConvDesc)]),
    'lmh_conv2d_winograd_transform_weights': (c_i, [P(ConvDesc), c_f, c_f, c_i, c_f, c_f]),
    'lmh_conv2d_winograd_v_bytes': (c_sz, [P(ConvDesc)]),
    'lmh_winograd_u_bytes': (c_sz, [c_i, c_i]),
    'lmh_winograd_transform_weights_batch': (c_i, [P(WinoWeightJob), c_i, c_i, c_f]),
    'lmh_x3_weights_bytes': (c_sz, [c_i, c_i, c_i, c_i]),
    'lmh_x3_split_weights_batch': (c_i, [P(X3WeightJob), c_i, c_i, c_f]),
    'lmh_conv2d_fwd_x3w_supported': (c_i, [P(ConvDesc)]),
    'lmh_conv2d_fwd_x3w': (c_i, [P(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    'lmh_conv2d_bwd_data_x3w_supported': (c_i, [P(ConvDesc)]),
    'lmh_conv2d_bwd_data_x3w': (c_i, [P(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    'lmh_conv2d_fwd_winograd': (c_i, [P(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_conv2d_bwd_data_winograd': (c_i, [P(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_conv2d_bwd_weight_winograd_workspace_bytes': (c_sz, [P(ConvDesc)]),
    'lmh_conv2d_bwd_weight_winograd': (c_i, [P(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_conv2d_bwd_weight_workspace_bytes': (c_sz, [P(ConvDesc)]),
    'lmh_conv2d_bwd_weight': (c_i, [P(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_conv2d_kernel_id': (c_i, [P(ConvDesc), c_i]),
    'lmh_conv2d_force_config': (None, [c_i, c_i, c_i]),
    'lmh_conv_set_stagger': (c_i, [c_i]),
    'lmh_conv2d_force_wgrad_variant': (None, [c_i]),
    'lmh_conv2d_bwd_weight_fuses_colsum': (c_i, [P(ConvDesc)]),
    'lmh_conv2d_profile_next': (c_i, [c_f, c_f]),
    'lmh_conv2d_profile_last': (ctypes.c_char_p, [P(ctypes.c_double)]),
    'lmh_conv2d_profile_last_bytes': (ctypes.c_double, []),
    'lmh_event_create': (ctypes.c_void_p, []),
    'lmh_event_destroy': (None, [c_f]),
    'lmh_event_elapsed_ms': (ctypes.c_float, [c_f, c_f]),
    'lmh_event_pair_overhead_ms': (ctypes.c_float, [c_i, c_f]),
    'lmh_act_fwd': (c_i, [c_f, c_f, c_i, c_i64, c_f]),
    'lmh_act_bwd_workspace_bytes': (c_sz, [c_i64, c_i]),
    'lmh_act_bwd': (c_i, [c_f, c_f, c_i, c_i64, c_i, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_bn_param_grads_workspace_bytes': (c_sz, [c_i64, c_i]),
    'lmh_bn_param_grads': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_i, c_f, c_f, c_sz, c_f]),
    'lmh_maxpool_fwd': (c_i, [c_f] + [c_i] * 10 + [c_f, c_f]),
    'lmh_maxpool_bwd': (c_i, [c_f, c_f, c_f] + [c_i] * 10 + [c_f, c_f]),
    'lmh_stream_wait_stream': (c_i, [c_f, c_f]),
    'lmh_event_record': (c_i, [c_f, c_f]),
    'lmh_stream_wait_event': (c_i, [c_f, c_f]),
    'lmh_memset': (c_i, [c_f, c_i, c_sz, c_f]),
    'lmh_memcpy_d2d': (c_i, [c_f, c_f, c_sz, c_f]),
    'lmh_plan_begin': (c_i, []),
    'lmh_plan_end': (ctypes.c_void_p, []),
    'lmh_plan_abort': (None, []),
    'lmh_plan_destroy': (None, [c_f]),
    'lmh_plan_recording': (c_i, []),
    'lmh_plan_position': (c_i, []),
    'lmh_plan_size': (c_i, [c_f]),
    'lmh_plan_kernel_count': (c_i, [c_f, c_i, c_i]),
    'lmh_plan_run': (c_i, [c_f, c_i, c_i]),
    'lmh_stream_create_cu_mask': (ctypes.c_void_p, [c_i, c_i]),
    'lmh_stream_create_cu_range': (ctypes.c_void_p, [c_i, c_i, c_i]),
    'lmh_stream_destroy': (None, [c_f]),
    'lmh_bn_refresh': (c_i, [c_f, c_f, c_f, c_f, c_i64, c_f, c_f, c_f]),
    'lmh_bn_train_workspace_bytes': (c_sz, [c_i64, c_i]),
    'lmh_bn_train_fwd': (c_i, [c_f, c_i64, c_i, c_f, c_f, c_fl, c_fl, c_f, c_f, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_bn_train_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i64, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_bn_apply': (c_i, [c_f, c_i64, c_i, c_f, c_f, c_f, c_i, c_f, c_f]),
    'lmh_loss_sums': (c_i, [ctypes.POINTER(ctypes.c_void_p), c_i, c_f, c_f, c_f, c_f]),
    'lmh_conv2d_hs_supported': (c_i, [P(ConvDesc)]),
    'lmh_conv2d_fwd_hs': (c_i, [P(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_f]),
    'lmh_conv2d_bwd_data_hs': (c_i, [P(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_i, ctypes.c_float, c_f]),
    'lmh_conv2d_bwd_weight_hs': (c_i, [P(ConvDesc), c_f, c_f, ctypes.c_float, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_half_weights_batch': (c_i, [P(HalfWeightJob), c_i, c_i, c_f]),
    'lmh_cast_to_half': (c_i, [c_f, ctypes.c_int64, c_i, ctypes.c_float, c_f, c_f, c_i, c_f]),
    'lmh_cast_to_f32': (c_i, [c_f, ctypes.c_int64, ctypes.c_float, c_f, c_i, c_f]),
    'lmh_maxpool_fwd_hs': (c_i, [c_f, c_i] + [c_i] * 10 + [c_f, c_i, c_f]),
    'lmh_subsample_bwd_hs': (c_i, [c_f] + [c_i] * 7 + [c_f, c_f]),
    'lmh_resize_bilinear': (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_f]),
    'lmh_rpn_proposal_workspace_bytes': (c_sz, [P(RpnProposalDesc)]),
    'lmh_rpn_proposal': (c_i, [P(RpnProposalDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_sort_u64': (c_i, [c_f, c_i, c_i, c_f]),
    'lmh_nms_workspace_bytes': (c_sz, [c_i, c_i]),
    'lmh_nms': (c_i, [c_f, c_f, c_i, c_i, c_fl, c_i, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_rpn_target_workspace_bytes': (c_sz, [P(RpnTargetDesc)]),
    'lmh_rpn_target': (c_i, [P(RpnTargetDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_rcnn_target_workspace_bytes': (c_sz, [P(RcnnTargetDesc)]),
    'lmh_rcnn_target': (c_i, [P(RcnnTargetDesc)] + [c_f] * 13 + [c_sz, c_f]),
    'lmh_rcnn_proposal_workspace_bytes': (c_sz, [P(RcnnProposalDesc)]),
    'lmh_rcnn_proposal': (c_i, [P(RcnnProposalDesc)] + [c_f] * 9 + [c_sz, c_f]),
    'lmh_ssd_proposal_workspace_bytes': (c_sz, [P(RcnnProposalDesc)]),
    'lmh_ssd_proposal': (c_i, [P(RcnnProposalDesc)] + [c_f] * 12 + [c_sz, c_f]),
    'lmh_roi_pool_fwd': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_fl, c_i, c_i, c_f, c_f, c_f]),
    'lmh_roi_pool_bwd_workspace_bytes': (c_sz, [c_i, c_i, c_i, c_i]),
    'lmh_roi_pool_bwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_fl, c_i, c_i, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_roi_pool_mean_supported': (c_i, [c_i, c_i, c_i]),
    'lmh_roi_pool_mean_fwd': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_fl, c_i, c_i, c_f, c_f, c_f]),
    'lmh_roi_pool_mean_bwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_fl, c_i, c_i, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_spatial_mean_fwd': (c_i, [c_f, c_i64, c_i, c_i, c_f, c_f]),
    'lmh_spatial_mean_bwd': (c_i, [c_f, c_i64, c_i, c_i, c_f, c_f]),
    'lmh_rpn_loss': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_fl, c_fl, c_fl, c_f, c_f, c_f, c_f, c_f]),
    'lmh_rcnn_loss': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_fl, c_fl, c_fl, c_f, c_f, c_f, c_f, c_f]),
    'lmh_rcnn_loss_grad': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_fl, c_fl, c_fl, c_f, c_f, c_f]),
    'lmh_softmax': (c_i, [c_f, c_i64, c_i, c_f, c_f]),
    'lmh_l2norm_scale_fwd': (c_i, [c_f, c_f, c_i64, c_i, c_fl, c_f, c_f]),
    'lmh_l2norm_scale_bwd_workspace_bytes': (c_sz, [c_i64, c_i]),
    'lmh_l2norm_scale_bwd': (c_i, [c_f, c_f, c_f, c_i64, c_i, c_fl, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_ssd_target_workspace_bytes': (c_sz, [P(SsdTargetDesc)]),
    'lmh_ssd_target': (c_i, [P(SsdTargetDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_f]),
    'lmh_ssd_loss': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_fl, c_fl, c_f, c_f, c_f, c_f, c_f]),
    'lmh_sgd_momentum': (c_i, [c_f, c_f, c_f, c_i64, c_f, c_f, c_i, c_fl, c_fl, c_fl, c_f]),
    'lmh_sgd_momentum_range': (c_i, [c_f, c_f, c_f, c_i64, c_i64, c_i64, c_f, c_f, c_i, c_f, c_fl, c_fl, c_i, c_f]),
    'lmh_tail_defer': (None, [c_i]),
    'lmh_tail_last_plan': (None, [P(ctypes.c_void_p), P(c_i), P(ctypes.c_void_p), P(c_i)]),
    'lmh_wgrad_tail_batch_workspace_bytes': (c_sz, [P(WgradTail), c_i]),
    'lmh_wgrad_tail_batch': (c_i, [P(WgradTail), c_i, c_f, c_sz, c_f]),
    'lmh_grad_clip_workspace_bytes': (c_sz, [c_i]),
    'lmh_grad_clip_factors': (c_i, [c_f, c_f, c_i64, c_f, c_f, c_i, c_fl, c_fl, c_f, c_f, c_sz, c_f]),
    'lmh_optimizer_step': (c_i, [c_i, c_f, c_f, c_f, c_f, c_f, c_i64, c_f, c_f, c_f, c_i, c_fl, c_fl, c_fl, c_fl, c_fl, c_f]),
    'lmh_dropout': (c_i, [c_f, c_i64, c_fl, ctypes.c_uint32, c_f, c_f]),
    'lmh_l2_reg_workspace_bytes': (c_sz, []),
    'lmh_l2_reg_loss': (c_i, [c_f, c_i64, c_f, c_f, c_i, c_f, c_f, c_sz, c_f]),
}


class LuminothHipError(RuntimeError):
    pass


def build(force=False):
    """Compile csrc/*.hip for gfx950 into csrc/libluminoth_hip.so (in-tree).  build.sh recompiles an object when its source,
    any header of csrc/ or the C-ABI header is newer, and prints which translation units it compiled and which it reused;
    `force` (or LUMINOTH_AMD_REBUILD=1) removes every object first."""
    force = force or os.environ.get('LUMINOTH_AMD_REBUILD') == '1'
    script = os.path.join(_HERE, 'csrc', 'build.sh')
    if force:
        for f in os.listdir(os.path.join(_HERE, 'csrc')):
            if f.endswith('.o'):
                os.remove(os.path.join(_HERE, 'csrc', f))
    subprocess.check_call(['bash', script])
    return LIB_PATH


def load():
    """Load the library and bind every declared symbol.  Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own libamdhip64; loaded before this library, the dynamic linker binds ours to the same
    # runtime instance.  The other order puts two HIP runtimes into the process and ours sees no device ("no ROCm-capable
    # device is detected" at the first launch) — met when build() and smoke() ran in one process.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise LuminothHipError(
            'libluminoth_hip.so not found at %s — run `python -c "import __graft_entry__ as g; '
            'g.build()"` (there is no CPU fallback on the product path)' % LIB_PATH)
    # a library built with LMH_PROBES=1 (timing probes in the convolution kernels; lmh_conv_set_stagger's decomposition
    # bits give deliberately wrong results) must not be picked up silently by a later product run
    flags_file = os.path.join(os.path.dirname(LIB_PATH), '.build_flags')
    try:
        probe_build = '-DLMH_PROBES' in open(flags_file).read()
    except OSError:
        probe_build = False
    if probe_build and os.environ.get('LMH_PROBES', '0') in ('', '0'):
        import warnings
        warnings.warn('libluminoth_hip.so was built with LMH_PROBES=1 (timing probes); rebuilding the product library')
        build(force=True)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # the C library reads no environment variable; sweeps / ablations set LMH_<OPTION>=<int> and it is forwarded here
    for key, val in os.environ.items():
        if key.startswith('LMH_OPT_'):
            rc = lib.lmh_set_default_option(key[len('LMH_OPT_'):].lower().encode(), int(val))    # process default: every thread
            if rc != 0:
                raise LuminothHipError('unknown tuning option %s' % key)
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().lmh_last_error()
        raise LuminothHipError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))
