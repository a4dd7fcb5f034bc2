"""ctypes binding of librelnet_hip.so -- the C-ABI declared in include/relnet_hip.h.

The library is the product: there is no CPU or PyTorch fallback.  Importing this module
when the library is missing raises, and every wrapper raises `RelnetError` on a non-zero
return code with the message recorded by the library.
"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RELNET_LIB') or os.path.join(HERE, 'librelnet_hip.so')      # (RELNET_LIB: A/B against another build)

F32, BF16 = 0, 1


class RelnetError(RuntimeError):
    pass


_lib = None

_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float

_SIGNATURES = {
    'relnet_version': (C.c_int, []),
    'relnet_last_error': (C.c_char_p, []),
    'relnet_gemm_nt': (C.c_int, [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _i, _vp, _i,
                                 _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_geometry_bias': (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i,
                                       _i, _i, _vp]),
    'relnet_relation_attention': (C.c_int, [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _i, _l, _vp,
                                            _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp,
                                            _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    'relnet_relation_attention_kc': (C.c_int, [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _i, _l, _vp,
                                               _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp,
                                               _i, _i, _i, _i, _i, _f, _i, _i, _vp, _vp]),
    'relnet_relation_attention_fused': (C.c_int, [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _i, _i, _vp, _vp, _vp,
                                                  _vp, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l,
                                                  _i, _i, _i, _i, _i, _f, _vp]),
    'relnet_bottleneck_chain': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp]),
    'relnet_bottleneck_chain_proj': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp]),
    'relnet_conv3x3_c64': (C.c_int, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    'relnet_proposal_decode': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_topk_sort': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'relnet_nms_mask': (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    'relnet_nms_scan': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    '_nms': (None, [_vp, _vp, _vp, _i, _i, _f, _i]),
    'relnet_roi_pool_fwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    'relnet_roi_align_fwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _vp]),
    'relnet_roi_align_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _vp]),
    'relnet_roi_pool_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _l, _l, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_roi_pool_bwd_ex': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_roi_pool_fpn_bwd_ex': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_detect_head': (C.c_int, [_vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'relnet_detect_head_ex': (C.c_int, [_vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'relnet_class_nms': (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, C.c_double, _i, _i, _vp]),
    'relnet_class_nms_ex': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, C.c_double, _i, _i, _vp]),
    'relnet_class_nms_topk': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, C.c_double, _i, _i, _i, _vp]),
    'relnet_class_nms_hist_bins': (C.c_int, []),
    'relnet_bbox_overlaps': (C.c_int, [_vp, _vp, _vp, _i, _i, _vp]),
    'relnet_image_topk': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'relnet_conv2d_nhwc': (C.c_int, [_vp, _l, _l, _vp, _vp, _vp, _i, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_conv2d_nhwc_f32': (C.c_int, [_vp, _l, _l, _vp, _vp, _vp, _i, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_maxpool_nhwc_f32': (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_pack_w_frag': (C.c_int, [_vp, _l, _vp, _i, _i, _vp]),
    'relnet_conv2d_nhwc_wf': (C.c_int, [_vp, _l, _l, _vp, _vp, _vp, _vp, _i, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_deformable_im2col': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _l] + [_i] * 15 + [_vp]),
    'relnet_deformable_psroi_pool_fwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 9 + [_f, _f, _i, _i, _i, _vp]),
    'relnet_roi_pool_fpn_fwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_fpn_roi_dispatch': (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'relnet_fpn_roi_dispatch_ex': (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    'relnet_upsample2x_add': (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'relnet_softmax_output': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _l, _i, _l, _i, _f, _f, _vp]),
    'relnet_softmax_output_ex': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _l, _i, _l, _i, _f, _f, _l, _vp]),
    'relnet_smooth_l1_loss': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _l, _f, _f, _vp]),
    'relnet_nms_loss': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _l, _f, _f, _f, _vp]),
    'relnet_transpose_2d': (C.c_int, [_vp, _l, _l, _vp, _l, _l, _i, _i, _i, _i, _vp]),
    'relnet_relation_attention_bwd': (C.c_int, [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _vp, _l, _l,
                                                _vp, _l, _l, _vp, _vp, _l, _l, _vp, _l, _l, _vp, _vp, _vp, _vp, _vp,
                                                _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    'relnet_relation_attention_bwd_kc': (C.c_int, [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _vp, _l, _l,
                                                   _vp, _l, _l, _vp, _vp, _l, _l, _vp, _l, _l, _vp, _vp, _vp, _vp, _vp,
                                                   _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    'relnet_geometry_bias_bwd': (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'relnet_relu_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _l, _i, _vp]),
    'relnet_strided_scatter': (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_colsum_add': (C.c_int, [_vp, _l, _l, _i, _i, _vp, _vp]),
    'relnet_colsum_add_grouped': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'relnet_sgd_update': (C.c_int, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _vp]),
    'relnet_deformable_col2im': (C.c_int, [_vp, _l, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 13 + [_vp]),
    'relnet_deformable_psroi_pool_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 9 + [_f, _f, _i, _i, _i, _vp]),
    'relnet_roi_pool_fpn_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_wgrad_accumulate': (C.c_int, [_vp, _i, _l, _i, _vp, _vp, _vp]),
    'relnet_wgrad_grouped': (C.c_int, [_vp, _i, _vp, _vp]),
    'relnet_wgrad_workspace_bytes': (C.c_long, [_i]),
    'relnet_wgrad_debug_plain': (None, [_i]),
    'relnet_wgrad_tune': (None, [_i, _i, _i]),
    'relnet_wgrad_debug_tiles': (None, [_i]),
    'relnet_debug_tr_probe': (C.c_int, [_vp, _vp]),
    'relnet_gemm_force_tile': (None, [_i]),
    'relnet_gemm_get_forced_tile': (C.c_int, []),
    'relnet_gemm_force_nloop': (None, [_i]),
    'relnet_weight_relayout': (C.c_int, [_vp, _i, _i, _vp]),
    'relnet_weight_fragpack': (C.c_int, [_vp, _i, _i, _vp]),
    'relnet_relation_bwd_pack': (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'relnet_lnms_scatter_bwd': (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'relnet_reduce_scalar': (C.c_int, [_vp, _l, _f, _i, _vp, _vp]),
    'relnet_lnms_pad_params': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'relnet_lnms_residual_relu': (C.c_int, [_vp, _vp, _vp, _l, _vp]),
    'relnet_lnms_cond_multi': (C.c_int, [_vp, _l, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'relnet_lnms_cond_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'relnet_lnms_take_bwd': (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'relnet_lnms_softmax_bwd': (C.c_int, [_vp, _vp, _vp, _l, _l, _i, _i, _i, _vp]),
    'relnet_lnms_gather_bias': (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_gemm_nt_mask': (C.c_int, [_vp, _l, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _vp]),
    'relnet_gemm_nt_f16': (C.c_int, [_vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _i, _vp]),
    'relnet_gemm_set_swizzle': (None, [_i]),
    'relnet_gemm_debug_korder': (None, [_i]),
    'relnet_gemm_debug_asm': (None, [_i]),
    'relnet_gemm_debug_phase_ts': (None, [_vp]),
    'relnet_gemm_debug_ablate': (None, [_i]),
    'relnet_chain_debug': (None, [_i]),
    'relnet_gemm_tile_count': (C.c_int, []),
    'relnet_gemm_set_workspace': (C.c_int, [_vp, _l]),
    'relnet_roi_pool_bwd_cl': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_roi_pool_bwd_debug': (None, [_i]),
    'relnet_deformable_col2im_debug': (None, [_i]),
    'relnet_relation_attention_debug_lds_f32': (None, [_i]),
    'relnet_deformable_psroi_pool_bwd_debug': (None, [_i]),
    'relnet_stream_capture_id': (C.c_ulonglong, [_vp]),
    'relnet_gemm_debug_splitk': (None, [_i]),
    'relnet_gemm_pick_tile': (C.c_int, [_i, _i, _i, _i, _i]),
    'relnet_nms_greedy': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    'relnet_stem_bias_relu_pool': (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_lnms_prepare': (C.c_int, [_vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'relnet_lnms_prepare_ex': (C.c_int, [_vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'relnet_lnms_sort': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'relnet_lnms_embed': (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_lnms_score': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _vp]),
    'relnet_stem_fused': (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'relnet_stem_pack_input': (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_stem_conv7': (C.c_int, [_vp, _vp, _vp, _i, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'relnet_proposal_target': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    'relnet_proposal_target_ex': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    'relnet_assign_anchor': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i,
                                       C.c_double, C.c_double, _i, _i, C.c_ulonglong, _vp, _vp]),
    'relnet_box_annotator_ohem': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'relnet_nms_multi_target': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
}


class RelayoutDesc(C.Structure):
    """relnet_relayout_desc (include/relnet_hip.h)."""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('cout', C.c_int), ('cin', C.c_int), ('taps', C.c_int), ('dst_ld', C.c_int),
                ('dst_co', C.c_int), ('tiles_co', C.c_int), ('tiles_ci', C.c_int), ('tile_start', C.c_int)]


class FragPackDesc(C.Structure):
    """relnet_fragpack_desc (include/relnet_hip.h)."""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('ldw', C.c_long), ('N', C.c_int), ('K', C.c_int), ('mode', C.c_int),
                ('block_start', C.c_int)]


class WgradDesc(C.Structure):
    """`relnet_wgrad_desc` of include/relnet_hip.h (one layer of a grouped weight-gradient launch)."""
    _fields_ = [('dy', C.c_void_p), ('dy_ld', C.c_long), ('dy_cols', C.c_int), ('x', C.c_void_p), ('x_pix', C.c_long),
                ('dw', C.c_void_p), ('dw_ld', C.c_long), ('row_scale', C.c_void_p)] + \
               [(n, C.c_int) for n in ('P', 'Cout', 'Cin', 'ks', 'stride', 'dil', 'pad', 'B', 'Hout', 'Wout', 'Hin', 'Win')]


def load():
    """dlopen the library (once) and attach argument types to every exported symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RelnetError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback path." % LIB_PATH)
    # torch FIRST: its wheel ships its own libamdhip64 / librocm runtime, and this library must bind to THAT copy.  dlopen'ed before torch is
    # imported (e.g. `build(); smoke()` in one process), it pulls in /opt/rocm's copy instead; torch then initialises a second HIP runtime and
    # every launch from here fails with "no ROCm-capable device is detected" -- the device memory handed in belongs to the other runtime.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale -> loud
        fn.restype = res
        fn.argtypes = args
    # A/B knobs for whole-step measurements without editing code.  They change which kernels run, so they are honoured only
    # together with RELNET_DEBUG_KNOBS=1 and every use is announced on stderr (a forced tile is a measured-slower configuration)
    knobs = [(k, os.environ[k]) for k in ('RELNET_GEMM_KORDER', 'RELNET_GEMM_FORCE_TILE', 'RELNET_GEMM_ASM', 'RELNET_GEMM_SPLITK', 'RELNET_WGRAD_TILES', 'RELNET_ROI_BWD_SCATTER', 'RELNET_ATTN_LDS_F32') if os.environ.get(k)]
    if knobs and os.environ.get('RELNET_DEBUG_KNOBS') != '1':
        sys.stderr.write('relnet: ignoring %s (set RELNET_DEBUG_KNOBS=1 to apply kernel-selection knobs)\n' % ', '.join(k for k, _ in knobs))
    elif knobs:
        sys.stderr.write('relnet: DEBUG kernel-selection knobs in effect: %s\n' % ', '.join('%s=%s' % kv for kv in knobs))
        for k, v in knobs:
            {'RELNET_GEMM_KORDER': lib.relnet_gemm_debug_korder, 'RELNET_GEMM_FORCE_TILE': lib.relnet_gemm_force_tile,
             'RELNET_GEMM_ASM': lib.relnet_gemm_debug_asm, 'RELNET_GEMM_SPLITK': lib.relnet_gemm_debug_splitk, 'RELNET_WGRAD_TILES': lib.relnet_wgrad_debug_tiles, 'RELNET_ROI_BWD_SCATTER': lib.relnet_roi_pool_bwd_debug, 'RELNET_ATTN_LDS_F32': lib.relnet_relation_attention_debug_lds_f32}[k](int(v))
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


#: optional per-launch timing: set to a callable(name) -> context manager (bench.py brackets
#: every C-ABI launch with HIP events on the launching stream); None = no overhead.
timing_hook = None


def call(name, *args, tag=None):
    lib = load()
    if timing_hook is not None:
        with timing_hook(name if tag is None else name + ':' + tag):
            rc = getattr(lib, name)(*args)
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RelnetError("%s failed (%d): %s" % (name, rc, lib.relnet_last_error().decode()))
