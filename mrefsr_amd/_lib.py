"""Loader for libmrefsr_hip.so (the C ABI of include/mrefsr_hip.h).  Fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MREFSR_HIP_LIB: another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get('MREFSR_HIP_LIB') or os.path.join(_HERE, 'lib', 'libmrefsr_hip.so')
ABI_VERSION = 1

_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64


class DcnShape(C.Structure):
    """mirror of mrefsr_dcn_shape"""
    _fields_ = [(n, C.c_int) for n in ('B', 'C', 'H', 'W', 'Co', 'kh', 'kw', 'stride_h', 'stride_w', 'pad_h', 'pad_w',
                                       'dil_h', 'dil_w', 'groups', 'dg')]


class ConvDesc(C.Structure):
    """mirror of mrefsr_conv_desc"""
    _fields_ = [(n, C.c_int32) for n in ('N', 'H', 'W', 'ksize', 'C1', 'ld1', 'N1', 'C2', 'ld2', 'N2', 'Cout', 'ld_out', 'ld_res',
                                         'pre_N', 'act', 'epilogue', 'terms')] + [('slope', C.c_float), ('wscale', C.c_float)]


class ConvPackJob(C.Structure):
    """mirror of mrefsr_conv_pack_job"""
    _fields_ = [('weight', C.c_void_p), ('packed', C.c_void_p), ('stride_o', C.c_int64), ('stride_i', C.c_int64), ('Cout', C.c_int32),
                ('Cin', C.c_int32), ('ksize', C.c_int32), ('terms', C.c_int32), ('flip', C.c_int32), ('wscale', C.c_float)]


# name -> (restype, argtypes): exactly the declarations of include/mrefsr_hip.h
SIGNATURES = {
    'mrefsr_abi_version': (_i, []),
    'mrefsr_last_error': (C.c_char_p, []),
    'mrefsr_corr_padded_channels': (_i, [_i]),
    'mrefsr_pixnorm_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'mrefsr_patch_norm_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'mrefsr_corr_top1_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'mrefsr_feature_match_index_workspace_bytes': (_i64, [_i, _i, _i, _i]),
    'mrefsr_feature_match_index_f32': (_i, [_vp, _vp] + [_i] * 10 + [_vp, _vp, _vp, _i64, _vp]),
    'mrefsr_corr_workspace_bytes': (_i64, [_i, _i, _i]),
    'mrefsr_corr_prefilter_info': (_i64, [_i, _i, _i, _i, C.c_char_p, _i, C.POINTER(C.c_int)]),
    'mrefsr_corr_top1_prefilter_f32': (_i, [_vp] * 9 + [_i64, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'mrefsr_offsets_from_idx_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'mrefsr_dynagg_prep_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'mrefsr_dynagg_prep_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mrefsr_dynagg_prep_bwd_nhwc_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mrefsr_dcn_fwd_workspace_bytes': (_i64, [C.POINTER(DcnShape)]),
    'mrefsr_dcn_fwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(DcnShape), _f, _i, _vp, _i64, _vp, _vp]),
    'mrefsr_dcn_fwd_amax_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(DcnShape), _f, _i, _vp, _i64, _vp, _vp, _vp]),
    'mrefsr_dcn_im2col_f32': (_i, [_vp, _vp, _vp, _vp, C.POINTER(DcnShape), _vp]),
    'mrefsr_dcn_col2im_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(DcnShape), _vp]),
    'mrefsr_dcn_bwd_data_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, C.POINTER(DcnShape), _vp]),
    'mrefsr_dcn_bwd_weight_workspace_bytes': (_i64, [C.POINTER(DcnShape)]),
    'mrefsr_conv_wgrad1x1_workspace_bytes': (_i64, [_i64, _i, _i]),
    'mrefsr_conv_wgrad1x1_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _i, _i, _i, _vp, _vp]),
    'mrefsr_dcn_bwd_weight_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(DcnShape), _vp, _vp]),
    'mrefsr_mrattn_fwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'mrefsr_mrattn_fwd_nhwc_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mrefsr_mrattn_fwd_nhwc_scaled_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    'mrefsr_mrattn_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'mrefsr_fused_bias_act': (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _f, _f, _i, _vp]),
    'mrefsr_bias_act_res_f32': (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i, _i64, _f, _vp]),
    'mrefsr_tail_bilinear_add_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'mrefsr_conv_packed_bytes': (_i64, [_i, _i, _i, _i]),
    'mrefsr_conv_pack_weight_f32': (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    'mrefsr_conv_pack_weight_view_f32': (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i64, _i64, _i, _vp]),
    'mrefsr_conv_pack_weights_multi_f32': (_i, [_vp, _i, _vp, _vp]),
    'mrefsr_act_bwd_blocks': (_i, [_i64, _i]),
    'mrefsr_act_bwd_nhwc_f32': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i64, _i, _i, _f, _vp, _vp, _vp]),
    'mrefsr_conv_nhwc_scaled_f32': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mrefsr_conv_nhwc_amax_f32': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mrefsr_conv_nhwc_bwd_f32': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mrefsr_conv_wgrad3x3_workspace_bytes': (_i64, [_i, _i, _i, _i, _i]),
    'mrefsr_conv_wgrad3x3_f32': (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i64, _i64, _i, _vp, _i, _i, _i, _vp, _i64, _vp, _vp]),
    'mrefsr_conv_wgrad3x3_batch_workspace_bytes': (_i64, [_i, _i, _i, _i, _i, _i]),
    'mrefsr_conv_wgrad3x3_batch_f32': (_i, [_i, _vp, _i, _i, _vp, _i, _i, _vp, _i64, _i64, _i, _vp, _i, _i, _i, _vp, _i64, _vp, _vp]),
    'mrefsr_mrattn_bwd_nhwc_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mrefsr_attn_modulate_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    'mrefsr_conv_nhwc_f32': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mrefsr_dcn_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(DcnShape), _f, _i, _vp]),
    'mrefsr_dcn_im2col': (_i, [_vp, _vp, _vp, _vp, C.POINTER(DcnShape), _i, _vp]),
    'mrefsr_dcn_col2im': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(DcnShape), _i, _vp]),
    'mrefsr_conv_dynagg_f32': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'mrefsr_mrattn_fwd_nhwc_bf16': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mrefsr_attn_modulate_bf16': (_i, [_vp, _vp, _vp, _i64, _vp]),
    'mrefsr_attn_modulate_f32': (_i, [_vp, _vp, _vp, _i64, _vp]),
    'mrefsr_bias_relu_pool2_f32': (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    'mrefsr_image_to_nhwc4_f32': (_i, [_vp, _vp, _i64, _i64, _i, _vp, _vp, _vp]),
    'mrefsr_weights_checksum': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    'mrefsr_upfirdn2d_f32': (_i, [_vp, _vp, _vp] + [_i] * 14 + [_vp]),
    'mrefsr_upfirdn2d': (_i, [_vp, _vp, _vp] + [_i] * 15 + [_vp]),
}

_lib = None


class MrefsrHipError(RuntimeError):
    pass


def load():
    """dlopen the library and bind every symbol of the header.  Needs no GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get('BASICSR_JIT') == 'True' and not os.environ.get('MREFSR_HIP_LIB'):
        # the reference's run-time build (basicsr/ops/dcn/deform_conv.py:10-21, fused_act.py:9-19, upfirdn2d.py:9-19:
        # torch.utils.cpp_extension.load of the CUDA sources at import), retargeted at hipcc: the library's own Makefile, incremental
        # (nothing to do when the sources are older than the .so), before the first dlopen.  BASICSR_EXT=True at `setup.py develop`
        # time (setup.py:118-136) corresponds to `make -C mrefsr_amd/csrc` / __graft_entry__.build() done once.
        import shutil
        import subprocess
        if shutil.which('hipcc') is None:
            raise MrefsrHipError('BASICSR_JIT=True asks for a run-time build of libmrefsr_hip.so, but hipcc is not on PATH')
        # one builder at a time: under torchrun / DDP every rank imports at once, and concurrent makes in the same _obj/ and lib/
        # would leave a rank dlopen-ing a half-written library (the reference's cpp_extension.load serialises with a file baton)
        import fcntl
        with open(os.path.join(_HERE, 'csrc', '.build.lock'), 'w') as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                subprocess.check_call(['make', '-C', os.path.join(_HERE, 'csrc'), '-s'])
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    if not os.path.exists(LIB_PATH):
        raise MrefsrHipError(
            f'{LIB_PATH} is not built. Build it with `make -C mrefsr_amd/csrc` (or '
            '`python -c "import __graft_entry__ as g; g.build()"`). There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library mismatch
        fn.restype, fn.argtypes = res, args
    got = lib.mrefsr_abi_version()
    if got != ABI_VERSION:
        raise MrefsrHipError(f'libmrefsr_hip.so ABI {got} != expected {ABI_VERSION}: rebuild')
    _lib = lib
    return lib


_TRACE = os.environ.get('MREFSR_TRACE_CALLS', '0') == '1'   # debugging: name every entry point on stderr before it runs (with
                                                            # AMD_SERIALIZE_KERNEL=3 the last name printed is the faulting launch)


def call(name, *args):
    lib = load()
    if _TRACE:
        import sys

        def show(a):
            o = getattr(a, '_obj', None)
            if isinstance(o, C.Structure):
                return '{' + ' '.join(f'{f}={getattr(o, f)}' for f, _ in o._fields_) + '}'
            return hex(a.value or 0) if isinstance(a, C.c_void_p) else str(getattr(a, 'value', a))
        print('[mrefsr]', name, ' '.join(show(a) for a in args), file=sys.stderr, flush=True)
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MrefsrHipError(f'{name} failed ({rc}): {lib.mrefsr_last_error().decode()}')
