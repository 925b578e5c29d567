"""ctypes binding of librpe_hip.so (the C ABI declared in include/rpe.h).

The product path has NO CPU or PyTorch fallback: if the library is missing or a call returns a non-zero
status, an exception is raised.  ``import torch`` must precede the dlopen so that the HIP runtime the
library binds to (SONAME libamdhip64.so.7) is the copy PyTorch-ROCm already loaded -- two HIP runtimes
in one process cannot share streams or device pointers.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (loads libamdhip64 first, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RPE_HIP_LIBRARY', os.path.join(_HERE, 'librpe_hip.so'))   # override: kernel experiments
CSRC = os.path.join(_HERE, 'csrc')

_c = ctypes
_vp, _i, _i64, _d, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_double, _c.c_size_t

_f, _ll = _c.c_void_p, _c.c_longlong


class ConvDesc(_c.Structure):
    """struct rpe_conv_desc (include/rpe.h)."""
    _fields_ = [('x', _f), ('x_batch_stride', _ll), ('packed', _f), ('bias', _f), ('add', _f), ('add_batch_stride', _ll),
                ('out', _f), ('out_batch_stride', _ll), ('out2', _f), ('out2_batch_stride', _ll),
                ('hidden', _f), ('hidden_batch_stride', _ll), ('zgate', _f), ('zgate_batch_stride', _ll),
                ('scale', _f), ('residual', _f), ('residual_batch_stride', _ll), ('stats', _f), ('pre_norm', _f),
                ('b', _i), ('cin', _i), ('cout', _i), ('h', _i), ('w', _i), ('kh', _i), ('kw', _i), ('mode', _i),
                ('gate_channels', _i), ('stride', _i), ('stats_tiles', _i)]


ABI_VERSION = 5            # RPE_ABI_VERSION of include/rpe.h these struct mirrors were written against


class SolveOpts(_c.Structure):
    """struct rpe_solve_opts (include/rpe.h)."""
    _fields_ = [('struct_size', _i), ('history_size', _i), ('tolerance_grad', _d), ('tolerance_change', _d), ('partition_rows', _i), ('reserved', _i)]


# name -> (restype, argtypes); mirrors include/rpe.h one to one
SIGNATURES = {
    'rpe_version': (_c.c_char_p, []),
    'rpe_abi_version': (_i, []),
    'rpe_se3_exp': (_i, [_vp, _vp, _i64, _i, _vp]),
    'rpe_se3_log': (_i, [_vp, _vp, _i64, _i, _vp]),
    'rpe_se3_mul': (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    'rpe_se3_inv': (_i, [_vp, _vp, _i64, _i, _vp]),
    'rpe_se3_act': (_i, [_vp, _vp, _vp, _i64, _i64, _i, _vp]),
    'rpe_se3_chain': (_i, [_vp, _vp, _vp, _i64, _d, _i, _vp]),
    'rpe_pose_gate_chain': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _d, _d, _i, _vp]),
    'rpe_pose_workspace_bytes': (_sz, [_i, _i, _i]),
    'rpe_pose_reduce': (_i, [_vp] * 10 + [_i, _i, _i, _i, _vp, _vp, _vp]),
    'rpe_pose_solve': (_i, [_vp] * 9 + [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'rpe_pose_solve_opts': (_i, [_vp] * 9 + [_i, _i, _i, _i, _i, _d, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'rpe_pose_solve_ex': (_i, [_vp] * 9 + [_i, _i, _i, _i, _i, _c.POINTER(SolveOpts), _vp, _vp, _vp, _vp, _vp, _vp]),
    'rpe_pose_backward_workspace_bytes': (_sz, [_i, _i, _i]),
    'rpe_pose_backward_moments': (_i, [_vp] * 10 + [_i, _i, _i, _vp, _vp, _vp]),
    'rpe_pose_backward_grads': (_i, [_vp] * 11 + [_i, _i, _i] + [_vp] * 6),
    'rpe_depth_backproject_warp': (_i, [_vp] * 9 + [_i, _i, _i] + [_vp] * 9),
    'rpe_flow2depth': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'rpe_warp_taps': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'rpe_corr_pyramid_bytes': (_sz, [_i, _i, _i, _i]),
    'rpe_corr_pyramid_bytes_ex': (_sz, [_i, _i, _i, _i, _i]),
    'rpe_corr_build': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'rpe_corr_build_ex': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'rpe_corr_lookup': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'rpe_corr_lookup_taps': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'rpe_corr_lookup_rounds': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'rpe_corr_export_level': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'rpe_gru_gates_zr': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    'rpe_gru_gates_h': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    'rpe_bias_act': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp]),
    'rpe_instnorm_act': (_i, [_vp, _vp, _i, _i, _i, _c.c_float, _i, _vp, _vp, _vp]),
    'rpe_affine_act': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'rpe_conv3x3_to2': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'rpe_conv3x3_to2_flow': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp]),
    'rpe_copy_planes': (_i, [_vp, _ll, _vp, _ll, _i, _i, _i, _vp]),
    'rpe_upsample_convex': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    'rpe_conv_packed_floats': (_sz, [_i, _i, _i, _i]),
    'rpe_conv_pack': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'rpe_conv1x1_packed_floats': (_sz, [_i, _i]),
    'rpe_conv1x1_pack': (_i, [_vp, _i, _i, _vp, _vp]),
    'rpe_conv_fused': (_i, [_c.POINTER(ConvDesc), _vp]),
    'rpe_conv_direct': (_i, [_vp, _ll, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _ll, _vp]),
    'rpe_conv_wino_packed_floats': (_sz, [_i, _i]),
    'rpe_conv_wino_stats_tiles': (_i, [_i, _i]),
    'rpe_conv_wino_pack': (_i, [_vp, _i, _i, _vp, _vp]),
    'rpe_conv_wino': (_i, [_c.POINTER(ConvDesc), _vp]),
    'rpe_conv_wino_x3_packed_bytes': (_sz, [_i, _i]),
    'rpe_conv_wino_x3_pack': (_i, [_vp, _i, _i, _vp, _vp]),
    'rpe_conv_wino_x3': (_i, [_c.POINTER(ConvDesc), _vp]),
    'rpe_conv1x1': (_i, [_c.POINTER(ConvDesc), _vp]),
    'rpe_conv1x1_x3_packed_bytes': (_sz, [_i, _i]),
    'rpe_conv1x1_x3_pack': (_i, [_vp, _i, _i, _vp, _vp]),
    'rpe_conv1x1_x3': (_i, [_c.POINTER(ConvDesc), _vp]),
    'rpe_conv_wino1d_packed_floats': (_sz, [_i, _i]),
    'rpe_conv_wino1d_pack': (_i, [_vp, _i, _i, _vp, _vp]),
    'rpe_conv_wino1d': (_i, [_c.POINTER(ConvDesc), _vp]),
    'rpe_conv_wino1d_x3_packed_bytes': (_sz, [_i, _i]),
    'rpe_conv_wino1d_x3_pack': (_i, [_vp, _i, _i, _vp, _vp]),
    'rpe_conv_wino1d_x3': (_i, [_c.POINTER(ConvDesc), _vp]),
    'rpe_conv_stats_tiles': (_i, [_i, _i, _i, _i]),
    'rpe_conv_stats_tiles_batch': (_i, [_i, _i, _i, _i, _i]),
    'rpe_instnorm_apply': (_i, [_vp, _vp, _i, _i, _i, _i, _c.c_float, _i, _vp, _vp, _vp]),
    'rpe_instnorm_apply_ex': (_i, [_vp, _vp, _i, _i, _i, _i, _c.c_float, _i, _vp, _vp, _vp, _vp]),
    'rpe_instnorm_finalize': (_i, [_vp, _i, _i, _i, _i, _c.c_float, _vp, _vp]),
    'rpe_unet_params_floats': (_sz, [_i]),
    'rpe_unet_workspace_bytes': (_sz, [_i, _i, _i]),
    'rpe_unet_heads': (_i, [_vp, _vp, _vp, _vp, _ll, _ll, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'rpe_stem_tiles': (_i, [_i, _i, _i]),
    'rpe_stem_packed_floats': (_sz, [_i, _i]),
    'rpe_stem_pack': (_i, [_vp, _i, _i, _vp, _vp]),
    'rpe_stem_conv': (_i, [_vp, _i, _i, _i, _i, _i, _c.c_float, _c.c_float, _c.c_float, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    'rpe_mask_specularities': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    'rpe_resize_crop': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'rpe_resize_crop_mask': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'rpe_remap_nearest': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    'rpe_shift_bilinear': (_i, [_vp, _i, _i, _i, _i, _c.c_float, _c.c_float, _vp, _vp]),
}

_lib = None


class RpeError(RuntimeError):
    pass


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into librpe_hip.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(['make', '-C', CSRC, '-j4'], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise RpeError('building librpe_hip.so failed')
    return LIB_PATH


def lib():
    """The loaded library, with argtypes set.  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RpeError(f'{LIB_PATH} is missing: run `python __graft_entry__.py` (build()) or `make -C {CSRC}`. '
                           'There is no CPU fallback for the HIP path.')
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.rpe_abi_version() != ABI_VERSION:
            raise RpeError(f'{LIB_PATH} has ABI version {L.rpe_abi_version()}, this binding expects {ABI_VERSION} (struct layouts of include/rpe.h): rebuild')
        _lib = L
    return _lib


_ERR = {-1: 'RPE_E_BADARG', -2: 'RPE_E_LAUNCH', -3: 'RPE_E_UNSUPPORTED'}


def check(status, what):
    if status != 0:
        raise RpeError(f'{what} failed with {_ERR.get(status, status)}')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr():
    """torch's current HIP stream of the current device as a void*.  (The raw accessor is ~10x cheaper than building a
    torch.cuda.Stream object per launch -- a frame of sequential tracking is ~330 launches.)"""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
