"""ctypes binding of the C ABI in include/transcar_hip.h.

The HIP library is the product; there is no CPU fallback.  ``lib()`` raises
``TransCARHipError`` if ``transcar_amd/lib/libtranscar_hip.so`` is missing
(build it with ``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C transcar_amd/csrc``), and every call checks the returned status.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
#: TRANSCAR_HIP_LIB selects another build of the same ABI.  The STAMPS debug build
#: (`make STAMPS=1`: s_memtime stamps and the TRANSCAR_CHAIN_DBG timing switches, which
#: produce WRONG results) is refused unless TRANSCAR_ALLOW_STAMPS=1 is set as well
#: (tools/chain_stamps.py); it is built under build/hip_stamps/, not into transcar_amd/lib/,
#: and is never the product.
LIB_PATH = os.environ.get('TRANSCAR_HIP_LIB') or os.path.join(_HERE, 'lib', 'libtranscar_hip.so')

TC_MAX_LEVELS = 4
TC_MAX_LAYERS = 8
TC_MAX_RADAR_LAYERS = 3
TC_ABI_VERSION = 12

c_fp = C.c_void_p      # device pointers travel as integers


class TransCARHipError(RuntimeError):
    pass


TC_MAX_RADAR_CHANNELS = 8
TC_SQ_NORM_PARTIALS = 256


class tc_radar_frame_desc(C.Structure):
    _fields_ = [('chan_start', C.c_int * (TC_MAX_RADAR_CHANNELS + 1)), ('num_chan', C.c_int),
                ('radar_rot', C.c_double * (TC_MAX_RADAR_CHANNELS * 9)), ('lidar_rot', C.c_double * 9),
                ('point_range', C.c_double * 6)]


class tc_linear(C.Structure):
    _fields_ = [('w', c_fp), ('b', c_fp)]


class tc_lnorm(C.Structure):
    _fields_ = [('g', c_fp), ('b', c_fp)]


class tc_pos_encoder(C.Structure):
    _fields_ = [('l0', tc_linear), ('n1', tc_lnorm), ('l3', tc_linear),
                ('n4', tc_lnorm)]


class tc_cls_branch(C.Structure):
    _fields_ = [('l0', tc_linear), ('n1', tc_lnorm), ('l3', tc_linear),
                ('n4', tc_lnorm), ('l6', tc_linear)]


class tc_reg_branch(C.Structure):
    _fields_ = [('l0', tc_linear), ('l2', tc_linear), ('l4', tc_linear)]


class tc_mha(C.Structure):
    _fields_ = [('in_proj', tc_linear), ('out_proj', tc_linear)]


class tc_decoder_layer(C.Structure):
    _fields_ = [('self_attn', tc_mha), ('norm0', tc_lnorm),
                ('attention_weights', tc_linear), ('output_proj', tc_linear),
                ('position_encoder', tc_pos_encoder), ('norm1', tc_lnorm),
                ('ffn0', tc_linear), ('ffn1', tc_linear), ('norm2', tc_lnorm),
                ('reg', tc_reg_branch), ('packed16_delta', C.c_size_t)]


class tc_radar_layer(C.Structure):
    _fields_ = [('attn', tc_mha), ('norm2', tc_lnorm), ('linear1', tc_linear),
                ('linear2', tc_linear), ('norm3', tc_lnorm),
                ('final_cls', tc_cls_branch), ('final_reg', tc_reg_branch),
                ('radius_min', C.c_float), ('radius_max', C.c_float), ('packed16_delta', C.c_size_t)]


class tc_head_weights(C.Structure):
    _fields_ = [('abi_version', C.c_int),
                ('num_query', C.c_int), ('embed_dims', C.c_int),
                ('num_heads', C.c_int), ('ffn_dims', C.c_int),
                ('num_layers', C.c_int),
                ('num_cams', C.c_int), ('num_levels', C.c_int),
                ('num_classes', C.c_int), ('code_size', C.c_int),
                ('radar_in_dims', C.c_int), ('num_radar_layers', C.c_int),
                ('num_radar_tokens_ref', C.c_int),
                ('pc_range', C.c_float * 6),
                ('query_embedding', c_fp),
                ('reference_points', tc_linear),
                ('layers', tc_decoder_layer * TC_MAX_LAYERS),
                ('radar_position_encoder', tc_pos_encoder),
                ('radar_feat0', tc_linear), ('radar_feat2', tc_linear),
                ('radar_feat4', tc_linear),
                ('radar', tc_radar_layer * TC_MAX_RADAR_LAYERS),
                ('l0_init_reference', c_fp), ('l0_attn_out', c_fp), ('packed16_delta', C.c_size_t)]


class tc_feats_nhwc(C.Structure):
    _fields_ = [('num_levels', C.c_int),
                ('data', c_fp * TC_MAX_LEVELS),
                ('H', C.c_int * TC_MAX_LEVELS),
                ('W', C.c_int * TC_MAX_LEVELS)]


class tc_head_aux(C.Structure):
    _fields_ = [('inter_states', c_fp), ('init_reference', c_fp),
                ('inter_references', c_fp), ('radar_hit_counts', c_fp),
                ('last_box', c_fp), ('sample_pairs', c_fp)]


class tc_head_options(C.Structure):
    _fields_ = [('chain_tile_rows', C.c_int), ('unfused', C.c_int),
                ('last_level_cls_only', C.c_int), ('reuse_radar_kv', C.c_int),
                ('decoder_dropout_p', C.c_float), ('radar_row_order', C.c_int),
                ('dropout_seed', C.c_ulonglong), ('phase', C.c_int), ('matrix_path', C.c_int),
                ('dropout_seed_stride', C.c_ulonglong), ('range_status', c_fp), ('cam_pregather', C.c_int),
                ('cam_pregather_ws', c_fp), ('cam_pregather_bytes', C.c_size_t)]


TC_MATRIX_AUTO, TC_MATRIX_F32, TC_MATRIX_F16X2 = 0, 1, 2


_P = C.POINTER
_i, _f, _sz, _vp = C.c_int, C.c_float, C.c_size_t, C.c_void_p

#: name -> (restype, argtypes); mirrors include/transcar_hip.h one to one
SIGNATURES = {
    'tc_abi_version': (_i, []),
    'tc_last_error': (C.c_char_p, []),
    'tc_device_count': (_i, []),
    'tc_nchw_to_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'tc_nchw_to_nhwc_levels': (_i, [_P(_vp), _P(_vp), _i, _i, _i, _P(_i), _P(_i), _vp]),
    'tc_radar_build_tokens': (_i, [_vp, _vp, _P(_i), _i, _P(C.c_double), _P(C.c_double),
                                   _P(C.c_double), _vp, _i, _vp, _vp]),
    'tc_radar_build_tokens_batch': (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp]),
    'tc_linear_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'tc_add_layernorm_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'tc_refine_reference_fwd': (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    'tc_cam_sample_fuse_fwd': (_i, [_P(tc_feats_nhwc), _i, _i, _i, _i, _vp,
                                    _vp, _vp, _P(_f), _f, _f, _vp, _vp, _vp,
                                    _vp]),
    'tc_cross_atten_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'tc_cross_atten_fwd': (_i, [_P(tc_linear), _P(tc_linear),
                                _P(tc_pos_encoder), _P(tc_feats_nhwc), _i, _i,
                                _i, _i, _vp, _vp, _vp, _vp, _P(_f), _f, _f,
                                _vp, _vp, _sz, _vp]),
    'tc_self_attn_workspace_bytes': (_sz, [_i, _i, _i]),
    'tc_self_attn_fwd': (_i, [_P(tc_mha), _vp, _vp, _vp, _i, _i, _i, _i, _vp,
                              _sz, _vp]),
    'tc_decoder_layer_tail_fwd': (_i, [_P(tc_decoder_layer), _P(tc_linear),
                                       _P(tc_feats_nhwc), _i, _i, _i, _i, _vp,
                                       _vp, _vp, _vp, _vp, _P(_f), _f, _f, _vp,
                                       _vp, _vp, _vp, _i, _i, _vp]),
    'tc_sdpa_fwd': (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'tc_sdpa_f16x2_workspace_bytes': (_sz, [_i, _i, _i]),
    'tc_sdpa_fwd_f16x2': (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'tc_radar_xattn_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'tc_radar_gate_selfcheck': (_i, [_i, C.c_ulonglong, _vp, _vp]),
    'tc_rowops_selfcheck': (_i, [_i, C.c_ulonglong, _vp, _vp]),
    'tc_radar_gated_xattn_fwd': (_i, [_P(tc_mha), _vp, _vp, _vp, _i, _vp, _vp,
                                      _i, _i, _i, _i, _i, _i, _f, _f, _vp,
                                      _vp, _vp, _sz, _vp]),
    'tc_radar_fusion_fwd': (_i, [_P(tc_head_weights), _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                                 _vp, _vp, _vp, _P(tc_head_options), _vp, _sz, _vp]),
    'tc_box_decode_workspace_bytes': (_sz, [_i, _i, _i]),
    'tc_box_decode_topk': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _P(_f), _vp,
                                _vp, _vp, _vp, _vp, _sz, _vp]),
    'tc_box_decode_kept': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _P(_f), _f, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'tc_head_workspace_bytes': (_sz, [_P(tc_head_weights), _i, _i]),
    'tc_cam_pregather_workspace_bytes': (_sz, [_P(tc_head_weights), _i]),
    'tc_head_packed_bytes': (_sz, [_P(tc_head_weights)]),
    'tc_head_pack_weights': (_i, [_P(tc_head_weights), _vp, _sz,
                                  _P(tc_head_weights), _vp]),
    'tc_head_repack_trainable': (_i, [_P(tc_head_weights), _P(tc_head_weights), _vp]),
    'tc_head_forward': (_i, [_P(tc_head_weights), _P(tc_head_weights),
                             _P(tc_feats_nhwc), _i, _vp,
                             _f, _f, _vp, _i, _i, _vp, _vp, _P(tc_head_aux),
                             _P(tc_head_options), _vp, _sz, _vp]),
    # training (backward of the trainable radar stack + optimizer)
    'tc_linear_gated_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'tc_linear_bwd_data': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f,
                                _i, _vp]),
    'tc_linear_bwd_weight': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f,
                                  _vp]),
    'tc_add_layernorm_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                  _i, _vp]),
    'tc_radar_reference_l1': (_i, [_vp, _P(_f), _vp, _vp, _i, _vp]),
    'tc_box_add_ref_fwd': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp]),
    'tc_box_add_ref_bwd': (_i, [_vp, _i, _vp, _i, _vp]),
    'tc_radar_attn_core_fwd': (_i, [_vp, _f, _vp, _vp, _i, _vp, _i, _vp, _i,
                                    _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp,
                                    _f, C.c_ulonglong, _i, _vp]),
    'tc_dropout': (_i, [_vp, _i, _i, _f, C.c_ulonglong, _i, _vp, _vp]),
    'tc_radar_attn_core_bwd': (_i, [_vp, _f, _vp, _vp, _i, _vp, _i, _vp, _i,
                                    _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp,
                                    _vp, _vp, _f, C.c_ulonglong, _i, _vp]),
    'tc_radar_train_tape_bytes': (_sz, [_P(tc_head_weights), _i, _i]),
    'tc_radar_train_fwd': (_i, [_P(tc_head_weights), _vp, _vp, _vp, _vp, _i, _i,
                                _i, _vp, _vp, _vp, _sz, _f, C.c_ulonglong, _vp]),
    'tc_radar_train_fwd_fused': (_i, [_P(tc_head_weights), _vp, _vp, _vp, _vp, _i, _i,
                                      _i, _vp, _vp, _vp, _sz, _f, C.c_ulonglong, _vp]),
    'tc_head_repack_trainable_ex': (_i, [_P(tc_head_weights), _P(tc_head_weights), _i, _vp]),
    'tc_radar_train_bwd': (_i, [_P(tc_head_weights), _P(tc_head_weights), _vp,
                                _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz,
                                _f, C.c_ulonglong, _vp]),
    'tc_radar_train_bwd_workspace_bytes': (_sz, [_P(tc_head_weights), _i, _i]),
    'tc_radar_train_bwd_fused': (_i, [_P(tc_head_weights), _P(tc_head_weights), _vp,
                                      _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp, _sz,
                                      _f, C.c_ulonglong, _vp, _vp, _vp]),
    'tc_radar_train_bwd_fused_ex': (_i, [_P(tc_head_weights), _P(tc_head_weights), _vp,
                                         _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp, _sz,
                                         _f, C.c_ulonglong, _vp, _vp, _i, _vp]),
    'tc_radar_train_bwd_fused_det': (_i, [_P(tc_head_weights), _P(tc_head_weights), _vp,
                                          _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp, _sz,
                                          _f, C.c_ulonglong, _vp, _vp, _i, _vp, _sz, _vp, _sz, _vp]),
    'tc_radar_train_bwd_weights': (_i, [_P(tc_head_weights), _P(tc_head_weights), _vp, _vp, _i, _i, _vp, _sz, _vp, _sz,
                                        _i, _vp]),
    'tc_radar_train_repack': (_i, [_P(tc_head_weights), _P(tc_head_weights), _vp, _sz, _i, _i, _vp]),
    'tc_dropout_mask': (_i, [_f, C.c_ulonglong, _i, _sz, _vp, _vp]),
    'tc_normalize_bbox': (_i, [_vp, _i, _vp, _vp]),
    'tc_match_cost': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _f, _f,
                           _f, _f, _f, _vp, _vp]),
    'tc_detr_loss_fwd_bwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp,
                                  _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    'tc_detr_loss_fwd_bwd_counts': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp,
                                         _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    'tc_lsa_assign': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'tc_lsa_assign_ex': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'tc_sq_norm': (_i, [_vp, _sz, _vp, _vp]),
    'tc_adamw_step': (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _f,
                           _f, _vp, _vp]),
}

_lib = None


def lib():
    """The loaded library (cached).  Fails loudly when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise TransCARHipError(
                'HIP library not built: %s is missing.  transcar_amd has no '
                'CPU fallback; run __graft_entry__.build() or '
                '`make -C transcar_amd/csrc`.' % LIB_PATH)
        dll = C.CDLL(LIB_PATH)
        if (hasattr(dll, 'tc_debug_chain_stamps') or hasattr(dll, 'tc_debug_diag_build')) and os.environ.get('TRANSCAR_ALLOW_STAMPS') != '1':
            raise TransCARHipError(
                '%s is the STAMPS debug build (timing switches with wrong results); set '
                'TRANSCAR_ALLOW_STAMPS=1 to use it for tools/chain_stamps.py' % LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(dll, name)        # AttributeError if a symbol is gone
            fn.restype = res
            fn.argtypes = args
        if dll.tc_abi_version() != TC_ABI_VERSION:
            raise TransCARHipError('ABI mismatch: library %d, binding %d' % (
                dll.tc_abi_version(), TC_ABI_VERSION))
        _lib = dll
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().tc_last_error().decode('utf-8', 'replace')
        raise TransCARHipError('%s failed (status %d): %s' % (what, rc, msg))


def f6(values):
    return (C.c_float * 6)(*[float(v) for v in values])
