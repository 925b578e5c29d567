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


class Op(_c.Structure):
    """struct rpe_op (include/rpe.h, prepared launch lists)."""
    _fields_ = [('kind', _i), ('stream', _i), ('args', _vp)]


_fl = _c.c_float


class CorrLookupArgs(_c.Structure):
    _fields_ = [('pyramid', _vp), ('coords', _vp), ('b', _i), ('h8', _i), ('w8', _i), ('levels', _i), ('radius', _i), ('out', _vp)]


class LookupConv1x1Args(_c.Structure):
    _fields_ = [('pyramid', _vp), ('coords', _vp), ('b', _i), ('h8', _i), ('w8', _i), ('levels', _i), ('radius', _i), ('packed', _vp), ('bias', _vp),
                ('relu', _i), ('out', _vp), ('out_batch_stride', _ll), ('out2', _vp), ('out2_batch_stride', _ll)]


class CorrBuildArgs(_c.Structure):
    _fields_ = [('fmap1', _vp), ('fmap2', _vp), ('b', _i), ('c', _i), ('h8', _i), ('w8', _i), ('levels', _i), ('feature_dtype', _i), ('pyramid', _vp)]


class StemConvArgs(_c.Structure):
    _fields_ = [('image', _vp), ('b', _i), ('cin', _i), ('h', _i), ('w', _i), ('stride', _i), ('div', _fl), ('mul', _fl), ('sub', _fl), ('packed', _vp),
                ('cout', _i), ('bias', _vp), ('scale', _vp), ('relu', _i), ('out', _vp), ('stats', _vp)]


class FlowUpdateArgs(_c.Structure):
    _fields_ = [('x', _vp), ('weight', _vp), ('bias', _vp), ('b', _i), ('c', _i), ('h', _i), ('w', _i), ('coords', _vp), ('coords_out', _vp),
                ('flow_out', _vp), ('dst1', _vp), ('dst1_batch_stride', _ll), ('dst2', _vp), ('dst2_batch_stride', _ll)]


class CopyPlanesArgs(_c.Structure):
    _fields_ = [('src', _vp), ('src_batch_stride', _ll), ('dst', _vp), ('dst_batch_stride', _ll), ('b', _i), ('c', _i), ('hw', _i)]


class InstnormFinalizeArgs(_c.Structure):
    _fields_ = [('partials', _vp), ('tiles', _i), ('b', _i), ('c', _i), ('hw', _i), ('eps', _fl), ('mean_inv', _vp)]


class InstnormApplyArgs(_c.Structure):
    _fields_ = [('x', _vp), ('partials', _vp), ('tiles', _i), ('b', _i), ('c', _i), ('hw', _i), ('eps', _fl), ('relu', _i), ('residual', _vp),
                ('residual_mean_inv', _vp), ('out', _vp)]


class UpsampleConvexArgs(_c.Structure):
    _fields_ = [('flow', _vp), ('mask', _vp), ('b', _i), ('h8', _i), ('w8', _i), ('out', _vp)]


# RPE_OP_* of include/rpe.h
OP_CONV_FUSED, OP_CONV_WINO, OP_CONV_WINO1D, OP_CONV1X1, OP_CONV_WINO_X3, OP_CONV_WINO1D_X3, OP_CONV1X1_X3 = 1, 2, 3, 4, 5, 6, 7
OP_CORR_LOOKUP, OP_STEM_CONV, OP_FLOW_UPDATE, OP_COPY_PLANES, OP_INSTNORM_FINALIZE, OP_INSTNORM_APPLY, OP_UPSAMPLE_CONVEX, OP_CORR_BUILD = 8, 9, 10, 11, 12, 13, 14, 15
OP_LOOKUP_CONV1X1 = 16
OP_EVENT_RECORD, OP_STREAM_WAIT = 32, 33
OP_OF_ENTRY = {'rpe_conv_fused': OP_CONV_FUSED, 'rpe_conv_wino': OP_CONV_WINO, 'rpe_conv_wino1d': OP_CONV_WINO1D, 'rpe_conv1x1': OP_CONV1X1,
               'rpe_conv_wino_x3': OP_CONV_WINO_X3, 'rpe_conv_wino1d_x3': OP_CONV_WINO1D_X3, 'rpe_conv1x1_x3': OP_CONV1X1_X3}

ABI_MINOR = 2              # RPE_ABI_MINOR: the newest additions this binding calls (rpe_run_ops)
ABI_VERSION = 5            # RPE_ABI_VERSION of include/rpe.h these struct mirrors were written against


class SolveOpts(_c.Structure):
    """struct rpe_solve_opts (include/rpe.h)."""
    _fields_ = [('struct_size', _i), ('history_size', _i), ('tolerance_grad', _d), ('tolerance_change', _d), ('partition_rows', _i), ('reserved', _i)]


# name -> (restype, argtypes); mirrors include/rpe.h one to one
SIGNATURES = {
    'rpe_version': (_c.c_char_p, []),
    'rpe_abi_version': (_i, []),
    'rpe_abi_minor': (_i, []),
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
    'rpe_corr_lookup_conv1x1_packed_floats': (_sz, [_i, _i]),
    'rpe_corr_lookup_conv1x1_pack': (_i, [_vp, _i, _i, _vp, _vp]),
    'rpe_corr_lookup_conv1x1': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _ll, _vp, _ll, _vp]),
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
    'rpe_run_ops': (_i, [_c.POINTER(Op), _i, _c.POINTER(_vp), _i, _c.POINTER(_i)]),
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
        if L.rpe_abi_minor() < ABI_MINOR:
            raise RpeError(f'{LIB_PATH} has ABI minor {L.rpe_abi_minor()}, this binding calls entry points added in minor {ABI_MINOR}: rebuild')
        _lib = L
    return _lib


class CountingLib:
    """Stands in for the loaded library (``with CountingLib() as c:``): counts the entry-point calls Python makes (``calls``; size queries
    are not launches) and the ops launch lists enqueue on top (``list_ops`` = rpe_run_ops' n_ops).  ops.Recorder uses it to prove that it
    logged every launch of the pass it recorded; bench.py reports the counts of a tracker frame."""

    def __init__(self):
        self._real, self.calls, self.list_ops, self.names = None, 0, 0, []

    def __enter__(self):
        global _lib
        self._real = lib()
        _lib = self
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self._real
        return False

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith('rpe_') or any(k in name for k in ('_bytes', '_floats', '_tiles', 'version')):        # (size queries launch nothing)
            return fn

        def counted(*a):
            self.calls += 1
            self.names.append(name)
            if name == 'rpe_run_ops':
                self.list_ops += a[1]
            return fn(*a)
        return counted


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


class RawEvents:
    """n raw hipEvent_t handles (timing enabled) from the HIP runtime this process already uses -- what a launch list's event cells take
    (ops.OpList; torch.cuda.Event creates its handle lazily and cannot be handed over before its first record).  Measurement plumbing
    for bench.py and the tests: ``handles`` (ints), ``elapsed_ms(i, j)``; the events are destroyed with the object."""

    def __init__(self, n):
        self._hip = None
        for line in open('/proc/self/maps'):
            if 'libamdhip64' in line:
                self._hip = ctypes.CDLL(line.split()[-1])           # the copy torch loaded (a second runtime could not share streams)
                break
        if self._hip is None:
            raise RpeError('RawEvents: no HIP runtime is loaded in this process')
        self._hip.hipEventCreate.argtypes = [_c.POINTER(_vp)]
        self._hip.hipEventElapsedTime.argtypes = [_c.POINTER(_c.c_float), _vp, _vp]
        self._hip.hipEventDestroy.argtypes = [_vp]
        self.handles = []
        for _ in range(n):
            h = _vp()
            if self._hip.hipEventCreate(_c.byref(h)) != 0:
                raise RpeError('hipEventCreate failed')
            self.handles.append(h.value)

    def elapsed_ms(self, i, j):
        ms = _c.c_float()
        st = self._hip.hipEventElapsedTime(_c.byref(ms), self.handles[i], self.handles[j])
        if st != 0:
            raise RpeError(f'hipEventElapsedTime failed with {st} (events not recorded / not complete?)')
        return ms.value

    def __del__(self):
        for h in getattr(self, 'handles', []):
            self._hip.hipEventDestroy(h)
        self.handles = []
