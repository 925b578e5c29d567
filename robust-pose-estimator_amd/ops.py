"""Tensor-level wrappers of the C ABI (include/rpe.h): argument checking, output allocation from torch's
caching allocator, and launch on torch's current HIP stream.  PyTorch is plumbing here (device memory and
streams); every operation below runs in librpe_hip.so.  Tensors must live on a ROCm device -- there is no
CPU path."""
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

SOLVER_LBFGS, SOLVER_GN = 0, 1


class OpList:
    """A prepared launch list for rpe_run_ops (include/rpe.h): the launches of a host loop -- RAFT.forward's update iterations, core/RAFT/
    core/raft.py -- enqueued by ONE call into the library instead of one Python-dispatched call each.  Built from the ``prepare=True``
    launchers of this module (their ``.op`` = (RPE_OP_* kind, argument struct)); every op runs through the same public entry point with
    the same arguments as the launcher would, so results are bit-identical to launching them one by one.  ``stream`` indexes the
    stream tuple given to run().  Event cells: cell(i) is a void* slot holding a raw hipEvent_t handle (0 = the op is skipped)."""

    def __init__(self, n_cells=0):
        import ctypes
        self._items, self._keep, self._arr = [], [], None
        self.cells = (ctypes.c_void_p * max(1, n_cells))()
        self._streams = (ctypes.c_void_p * 4)()
        self._failed = ctypes.c_int(-1)

    def __len__(self):
        return len(self._items)

    def add(self, launcher, stream=0):
        kind, args = launcher.op
        import ctypes
        self._items.append((kind, stream, ctypes.addressof(args)))
        self._keep.append(launcher)                       # (the launcher keeps the struct and every tensor it points to alive)
        self._arr = None
        return self

    def _cell_op(self, kind, cell, stream):
        import ctypes
        self._items.append((kind, stream, ctypes.addressof(self.cells) + ctypes.sizeof(ctypes.c_void_p) * cell))
        self._arr = None
        return self

    def record(self, cell, stream=0):
        return self._cell_op(_lib.OP_EVENT_RECORD, cell, stream)

    def wait(self, cell, stream=0):
        return self._cell_op(_lib.OP_STREAM_WAIT, cell, stream)

    def mark(self):
        """Index of the next op: run(start, stop) takes such marks (a caller that needs the state between iterations runs slices)."""
        return len(self._items)

    def run(self, streams, start=0, stop=None):
        """Enqueue ops [start, stop) on ``streams`` (raw hipStream_t handles as ints, index = the ops' ``stream``)."""
        import ctypes
        if self._arr is None:
            self._arr = (_lib.Op * max(1, len(self._items)))(*[_lib.Op(k, s, a) for k, s, a in self._items])
        stop = len(self._items) if stop is None else stop
        if stop <= start:
            return
        for i, h in enumerate(streams):
            self._streams[i] = h
        st = lib().rpe_run_ops(ctypes.cast(ctypes.byref(self._arr, ctypes.sizeof(_lib.Op) * start), ctypes.POINTER(_lib.Op)), stop - start, self._streams,
                               len(streams), ctypes.byref(self._failed))
        if st != 0:
            check(st, f'rpe_run_ops (op {start + self._failed.value} of the list, kind {self._items[start + self._failed.value][0]})')


_REC = None          # the active Recorder (one host thread drives one GPU)


class Recorder(OpList):
    """An OpList filled by RUNNING a piece of host code once: inside ``with recorder:`` every wrapper of this module that a launch list
    can carry (the convolutions, stems, instance-norm passes, plane copies, the correlation build, the convex up-sampling) launches as
    usual AND logs its argument block -- so the recording pass is an ordinary pass with an ordinary result.  Every tensor a logged
    launch touches is kept alive by the recorder: the intermediates of the pass become the list's private workspace, at fixed addresses.
    ``bind(name, tensor)`` then marks a tensor of the pass as an external input / output: replay({name: new_tensor, ...}) rewrites every
    pointer of the list that points into it (base + the same offset: channel / batch slices stay slices) and enqueues the whole list
    with one rpe_run_ops call on the current stream.  The caller guarantees what the list cannot see: same shapes and dtypes, same
    weights, replays on the stream it was recorded on (the workspace is reused without synchronisation)."""

    def __init__(self):
        super().__init__()
        self._structs, self._bound, self._claimed = [], {}, set()
        self.complete = False         # set when the pass has ended and every library launch of it was logged

    def __enter__(self):
        global _REC
        if _REC is not None:
            raise _lib.RpeError('Recorder: recordings do not nest')
        self._count = _lib.CountingLib().__enter__()
        _REC = self
        return self

    def __exit__(self, *exc):
        global _REC
        _REC = None
        self._count.__exit__(*exc)
        # a launch that went to the library without being logged (a wrapper this class does not know, a fallback route) would be
        # missing from every replay: such a recording is not usable, and the caller keeps launching the pass call by call
        self.complete = exc[0] is None and self._count.calls == len(self._items)
        self.unlogged = [] if self.complete else sorted(set(self._count.names))
        del self._count
        return False

    def log(self, kind, args, keep):
        import ctypes
        self._items.append((kind, 0, ctypes.addressof(args)))
        self._keep.append((args, keep))
        self._structs.append(args)
        self._arr = None

    def bind(self, name, t):
        """Every pointer field of the logged argument blocks that points into ``t``'s memory -> (struct, field, offset)."""
        import ctypes
        lo, hi = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()
        sites = []
        for i, st in enumerate(self._structs):
            for fname, ftype in st._fields_:
                if ftype is ctypes.c_void_p and (i, fname) not in self._claimed:
                    v = getattr(st, fname)
                    if v is not None and lo <= v < hi:
                        sites.append((st, fname, v - lo))
                        self._claimed.add((i, fname))
        self._bound[name] = (sites, tuple(t.shape), t.dtype)
        return len(sites)

    def patch(self, tensors):
        """Rewrite every pointer bound to ``name`` for the tensors given (same shape and dtype as at recording time, contiguous)."""
        for name, t in tensors.items():
            sites, shape, dtype = self._bound[name]
            if tuple(t.shape) != shape or t.dtype != dtype or not t.is_contiguous():
                raise _lib.RpeError(f'Recorder.replay: {name} must be a contiguous {dtype} tensor of shape {shape}')
            base = t.data_ptr()
            for st, fname, off in sites:
                setattr(st, fname, base + off)

    def replay(self, tensors):
        self.patch(tensors)
        self.run((raw_stream(),))


def raw_stream(stream=None):
    """The raw hipStream_t handle (int) of a torch stream (default: the current stream of the current device)."""
    return _lib.stream_ptr().value or 0 if stream is None else stream.cuda_stream
_DT = {torch.float32: 0, torch.float64: 1}


def _on_current_device(t, name):
    """Kernels are enqueued on the CURRENT device's stream: a tensor of another GPU would be written through a foreign
    pointer.  One process drives one GPU here (torch.cuda.set_device(LOCAL_RANK)); anything else is refused."""
    if t.device.index != torch.cuda.current_device():
        raise _lib.RpeError(f'{name}: tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()} '
                            '(call torch.cuda.set_device first; multi-device use from one process is unsupported)')


def _dev(t, dtype=None, name='tensor'):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.RpeError(f'{name}: expected a tensor on the GPU (the HIP path has no CPU fallback)')
    _on_current_device(t, name)
    if dtype is not None and t.dtype != dtype:
        raise _lib.RpeError(f'{name}: expected dtype {dtype}, got {t.dtype}')
    return t if t.is_contiguous() else t.contiguous()


def _mask(t, name):
    t = _dev(t, None, name)
    if t.dtype == torch.bool:
        return t.view(torch.uint8)
    if t.dtype == torch.uint8:
        return t
    raise _lib.RpeError(f'{name}: expected bool/uint8 mask')


# ------------------------------------------------------------------------------------------------- SE(3)
def _se3_unary(fn, x, din, dout):
    x = _dev(x, None, 'se3 input')
    if x.dtype not in _DT or x.shape[-1] != din:
        raise _lib.RpeError('se3: bad dtype/shape')
    out = torch.empty(*x.shape[:-1], dout, dtype=x.dtype, device=x.device)
    n = x.numel() // din
    check(fn(ptr(x), ptr(out), n, _DT[x.dtype], stream_ptr()), 'rpe_se3')
    return out


def se3_exp(xi):
    return _se3_unary(lib().rpe_se3_exp, xi, 6, 7)


def se3_log(T):
    return _se3_unary(lib().rpe_se3_log, T, 7, 6)


def se3_inv(T):
    return _se3_unary(lib().rpe_se3_inv, T, 7, 7)


def se3_mul(A, B):
    A, B = torch.broadcast_tensors(A, B)
    A, B = _dev(A, None, 'A'), _dev(B, A.dtype, 'B')
    out = torch.empty_like(A)
    check(lib().rpe_se3_mul(ptr(A), ptr(B), ptr(out), A.numel() // 7, _DT[A.dtype], stream_ptr()), 'rpe_se3_mul')
    return out


def se3_act(T, pts):
    """T (n,7) or (n,1,7) acting on pts (n,m,3)."""
    pts = _dev(pts, None, 'pts')
    n, m = pts.shape[0], pts.shape[1]
    T = _dev(T.reshape(n, 7), pts.dtype, 'T')
    out = torch.empty_like(pts)
    check(lib().rpe_se3_act(ptr(T), ptr(pts), ptr(out), n, m, _DT[pts.dtype], stream_ptr()), 'rpe_se3_act')
    return out


def se3_chain(rel, scale=1.0, init=None):
    rel = _dev(rel.reshape(-1, 7), None, 'rel')
    init = _dev(init.reshape(7), rel.dtype, 'init') if init is not None else None
    out = torch.empty_like(rel)
    check(lib().rpe_se3_chain(ptr(rel), ptr(init), ptr(out), rel.shape[0], float(scale), _DT[rel.dtype], stream_ptr()),
          'rpe_se3_chain')
    return out


def pose_gate_chain(rel, init, scale, thr=0.1):
    """rpe_pose_gate_chain: PoseEstimator's failure gate and pose chaining for m relative poses (m,7) in one launch.
    Returns (gated relative poses (m,7), absolute poses (m,7), ok (m,) int32)."""
    rel = _dev(rel.reshape(-1, 7), None, 'rel')
    init = _dev(init.reshape(7), rel.dtype, 'init') if init is not None else None
    rel_out, abs_out = torch.empty_like(rel), torch.empty_like(rel)
    ok = torch.empty(rel.shape[0], dtype=torch.int32, device=rel.device)
    check(lib().rpe_pose_gate_chain(ptr(rel), ptr(init), ptr(rel_out), ptr(abs_out), ptr(ok), rel.shape[0], float(scale), float(thr),
                                    _DT[rel.dtype], stream_ptr()), 'rpe_pose_gate_chain')
    return rel_out, abs_out, ok


# ------------------------------------------------------------------------------------------------- pose layer
def _pose_inputs(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight):
    f32 = torch.float32
    flow = _dev(flow, f32, 'flow')
    n, _, h, w = flow.shape
    pcl1, pcl2 = _dev(pcl1, f32, 'pcl1'), _dev(pcl2, f32, 'pcl2')
    w1, w2 = _dev(w1, f32, 'w1'), _dev(w2, f32, 'w2')
    mask1, mask2 = _mask(mask1, 'mask1'), _mask(mask2, 'mask2')
    K = _dev(K, f32, 'K')
    lw = _dev(loss_weight, f32, 'loss_weight')
    for t, c in ((pcl1, 3), (pcl2, 3), (w1, 1), (w2, 1), (mask1, 1), (mask2, 1)):
        if tuple(t.shape) != (n, c, h, w):
            raise _lib.RpeError(f'pose layer: shape {tuple(t.shape)} != {(n, c, h, w)}')
    if tuple(K.shape) != (n, 3, 3) or tuple(lw.shape) != (n, 2):
        raise _lib.RpeError('pose layer: K must be (n,3,3) and loss_weight (n,2)')
    return (flow, pcl1, pcl2, w1, w2, mask1, mask2, K, lw), n, h, w


def _workspace(n, h, w, device):
    return torch.empty(lib().rpe_pose_workspace_bytes(n, h, w), dtype=torch.uint8, device=device)


def pose_reduce(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, T, need_hessian=False):
    """One objective evaluation at T (n,7) f64 -> dict(loss2d, loss3d, f, g (n,6), H (n,6,6))."""
    args, n, h, w = _pose_inputs(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight)
    T = _dev(T.reshape(n, 7), torch.float64, 'T')
    out = torch.empty(n, 32, dtype=torch.float64, device=T.device)
    ws = _workspace(n, h, w, T.device)
    check(lib().rpe_pose_reduce(*[ptr(a) for a in args], ptr(T), n, h, w, int(bool(need_hessian)), ptr(out), ptr(ws),
                                stream_ptr()), 'rpe_pose_reduce')
    res = dict(loss2d=out[:, 0], loss3d=out[:, 1], f=out[:, 2], g=out[:, 3:9])
    if need_hessian:
        iu = torch.triu_indices(6, 6, device=T.device)
        H = torch.zeros(n, 6, 6, dtype=torch.float64, device=T.device)
        H[:, iu[0], iu[1]] = out[:, 9:30]
        H = H + H.transpose(1, 2) - torch.diag_embed(torch.diagonal(H, dim1=1, dim2=2))
        res['H'] = H
    return res


def pose_solve(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, iters, mode=SOLVER_LBFGS,
               tolerance_grad=1e-7, tolerance_change=1e-9, history_size=100, partition_rows=0, persistent=True):
    """Device-resident solve.  Returns (T f64 (n,7), vec7 f32 (n,7), log6 f32 (n,6), info int32 (n,4)).
    ``partition_rows=1``: every row's float64 sums are grouped as if it were solved alone (rpe_solve_opts), i.e. the result of a
    row does not depend on the batch it is in, bit for bit.  ``persistent=False`` (RPE_SOLVE_LAUNCH_PER_EVALUATION): one launch per
    evaluation instead of one for the whole solve -- the same bits, for A/B measurements."""
    args, n, h, w = _pose_inputs(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight)
    dev = args[0].device
    T = torch.empty(n, 7, dtype=torch.float64, device=dev)
    vec7 = torch.empty(n, 7, dtype=torch.float32, device=dev)
    log6 = torch.empty(n, 6, dtype=torch.float32, device=dev)
    info = torch.empty(n, 4, dtype=torch.int32, device=dev)
    ws = _workspace(n, h, w, dev)
    import ctypes
    o = _lib.SolveOpts(ctypes.sizeof(_lib.SolveOpts), int(history_size), float(tolerance_grad), float(tolerance_change), int(partition_rows),
                       0 if persistent else 1)
    check(lib().rpe_pose_solve_ex(*[ptr(a) for a in args], n, h, w, int(mode), int(iters), ctypes.byref(o), ptr(T), ptr(vec7), ptr(log6),
                                  ptr(info), ptr(ws), stream_ptr()), 'rpe_pose_solve_ex')
    return T, vec7, log6, info


def pose_backward_moments(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, T):
    """At pose T (n,7) f64: (g2u (n,6), g3u (n,6), H (n,6,6)) -- unit-loss-weight tangent gradients of the two terms
    and the symmetrised fYY of the reference's backward (declerative_node_lie.py:40-51)."""
    args, n, h, w = _pose_inputs(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight)
    T = _dev(T.reshape(n, 7), torch.float64, 'T')
    out = torch.empty(n, 48, dtype=torch.float64, device=T.device)
    ws = torch.empty(lib().rpe_pose_backward_workspace_bytes(n, h, w), dtype=torch.uint8, device=T.device)
    check(lib().rpe_pose_backward_moments(*[ptr(a) for a in args], ptr(T), n, h, w, ptr(out), ptr(ws), stream_ptr()),
          'rpe_pose_backward_moments')
    return out[:, :6], out[:, 6:12], out[:, 12:].reshape(n, 6, 6)


def pose_backward_grads(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight, T, u, want):
    """fXY^T u for the inputs named in ``want`` (subset of flow, pcl1, pcl2, w1, w2) -> dict of float32 tensors."""
    args, n, h, w = _pose_inputs(flow, pcl1, pcl2, w1, w2, mask1, mask2, K, loss_weight)
    T = _dev(T.reshape(n, 7), torch.float64, 'T')
    u = _dev(u.reshape(n, 6), torch.float64, 'u')
    ch = dict(flow=2, pcl1=3, pcl2=3, w1=1, w2=1)
    outs = {k: (torch.empty(n, c, h, w, dtype=torch.float32, device=T.device) if k in want else None) for k, c in ch.items()}
    check(lib().rpe_pose_backward_grads(*[ptr(a) for a in args], ptr(T), ptr(u), n, h, w, ptr(outs['flow']), ptr(outs['pcl1']),
                                        ptr(outs['pcl2']), ptr(outs['w1']), ptr(outs['w2']), stream_ptr()), 'rpe_pose_backward_grads')
    return {k: v for k, v in outs.items() if v is not None}


# ------------------------------------------------------------------------------------------------- geometry
def depth_backproject_warp(stereo_flow2, time_flow, baseline, K, depth1, image1l, image2l, stereo_flow1, mask2,
                           want_pcl2=False):
    f32 = torch.float32
    sf2 = _dev(stereo_flow2, f32, 'stereo_flow2')
    n, _, h, w = sf2.shape
    tf, b, K = _dev(time_flow, f32, 'time_flow'), _dev(baseline, f32, 'baseline'), _dev(K, f32, 'K')
    d1, i1, i2 = _dev(depth1, f32, 'depth1'), _dev(image1l, f32, 'image1l'), _dev(image2l, f32, 'image2l')
    sf1, m2 = _dev(stereo_flow1, f32, 'stereo_flow1'), _mask(mask2, 'mask2')
    dev = sf2.device
    e = lambda *s, dt=f32: torch.empty(*s, dtype=dt, device=dev)
    depth2, pcl1, pcl2w = e(n, 1, h, w), e(n, 3, h, w), e(n, 3, h, w)
    m2v, m2w = e(n, 1, h, w, dt=torch.uint8), e(n, 1, h, w, dt=torch.uint8)
    inp1, inp2 = e(n, 8, h // 8, w // 8), e(n, 8, h // 8, w // 8)
    pcl2 = e(n, 3, h, w) if want_pcl2 else None
    check(lib().rpe_depth_backproject_warp(ptr(sf2), ptr(tf), ptr(b), ptr(K), ptr(d1), ptr(i1), ptr(i2), ptr(sf1), ptr(m2),
                                           n, h, w, ptr(depth2), ptr(m2v), ptr(pcl1), ptr(pcl2w), ptr(m2w), ptr(inp1),
                                           ptr(inp2), ptr(pcl2), stream_ptr()), 'rpe_depth_backproject_warp')
    return dict(depth2=depth2, mask2=m2v.view(torch.bool), pcl1=pcl1, pcl2w=pcl2w, mask2w=m2w.view(torch.bool),
                inp1=inp1, inp2=inp2, pcl2=pcl2)


def flow2depth(stereo_flow, baseline):
    sf = _dev(stereo_flow, torch.float32, 'stereo_flow')
    n, _, h, w = sf.shape
    b = _dev(baseline, torch.float32, 'baseline')
    depth = torch.empty(n, 1, h, w, dtype=torch.float32, device=sf.device)
    valid = torch.empty(n, 1, h, w, dtype=torch.uint8, device=sf.device)
    check(lib().rpe_flow2depth(ptr(sf), ptr(b), n, h, w, ptr(depth), ptr(valid), stream_ptr()), 'rpe_flow2depth')
    return depth, valid.view(torch.bool)


def warp_taps(flow):
    fl = _dev(flow, torch.float32, 'flow')
    n, _, h, w = fl.shape
    outs = [torch.empty(n, h, w, dtype=torch.int32, device=fl.device) for _ in range(4)]
    check(lib().rpe_warp_taps(ptr(fl), n, h, w, *[ptr(o) for o in outs], stream_ptr()), 'rpe_warp_taps')
    return dict(x0=outs[0], y0=outs[1], xn=outs[2], yn=outs[3])


# ------------------------------------------------------------------------------------------------- correlation
class CorrPyramid:
    """Opaque device buffer holding the 4-level correlation pyramid of a batch of pairs."""

    def __init__(self, b, h8, w8, levels=4, radius=4, device='cuda', bf16x3=False):
        """``bf16x3``: size the scratch for the RPE_F32X3 experiment (1.5x the feature-map scratch); build(bf16x3=True) needs it."""
        self.b, self.h8, self.w8, self.levels, self.radius = b, h8, w8, levels, radius
        nbytes = lib().rpe_corr_pyramid_bytes_ex(b, h8, w8, levels, 3 if bf16x3 else 0)
        if nbytes == 0:
            raise _lib.RpeError('rpe_corr_pyramid_bytes: unsupported geometry')
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)

    def build(self, fmap1, fmap2, fp16_features=False, bf16x3=False):
        """``fp16_features``: BASELINE config 5 -- both maps are rounded to fp16 and correlated on the 16-bit matrix cores
        with f32 accumulation; the pyramid stays f32.  ``bf16x3``: f32 features, every f32 product evaluated as six bf16 products of
        an exact three-way split (RPE_F32X3: f32-equivalent results at 3/8 of the matrix time)."""
        f1, f2 = _dev(fmap1, torch.float32, 'fmap1'), _dev(fmap2, torch.float32, 'fmap2')
        b, c, h8, w8 = f1.shape
        if (b, h8, w8) != (self.b, self.h8, self.w8) or f2.shape != f1.shape:
            raise _lib.RpeError('corr build: shape mismatch')
        if self.buf.numel() < lib().rpe_corr_pyramid_bytes_ex(b, h8, w8, self.levels, 2 if fp16_features else 3 if bf16x3 else 0):
            raise _lib.RpeError('corr build: this pyramid was not sized for the requested feature mode (CorrPyramid(..., bf16x3=True))')
        if _REC is not None:
            _REC.log(_lib.OP_CORR_BUILD, _lib.CorrBuildArgs(f1.data_ptr(), f2.data_ptr(), b, c, h8, w8, self.levels, 2 if fp16_features else 3 if bf16x3 else 0,
                                                            self.buf.data_ptr()), (f1, f2, self))
        check(lib().rpe_corr_build_ex(ptr(f1), ptr(f2), b, c, h8, w8, self.levels, 2 if fp16_features else 3 if bf16x3 else 0, ptr(self.buf),
                                      stream_ptr()), 'rpe_corr_build_ex')
        return self

    def lookup(self, coords, out=None, prepare=False):
        """``prepare=True`` (needs ``out``): a zero-argument launcher on these buffers, with ``.op`` for an OpList."""
        co = _dev(coords, torch.float32, 'coords')
        if tuple(co.shape) != (self.b, 2, self.h8, self.w8):
            raise _lib.RpeError('corr lookup: coords shape mismatch')
        ch = self.levels * (2 * self.radius + 1) ** 2
        if out is None:
            out = torch.empty(self.b, ch, self.h8, self.w8, dtype=torch.float32, device=co.device)
        if prepare:
            if co is not coords or tuple(_nchw(out, 'out').shape) != (self.b, ch, self.h8, self.w8):
                raise _lib.RpeError('corr lookup: a prepared launch needs contiguous coords and a (b, levels*(2r+1)^2, h8, w8) out buffer')
            a = _lib.CorrLookupArgs(self.buf.data_ptr(), co.data_ptr(), self.b, self.h8, self.w8, self.levels, self.radius, out.data_ptr())
            fn, keep = lib().rpe_corr_lookup, (self, co, out)

            def launch():
                if _REC is not None:
                    _REC.log(_lib.OP_CORR_LOOKUP, a, keep)
                check(fn(a.pyramid, a.coords, a.b, a.h8, a.w8, a.levels, a.radius, a.out, stream_ptr()), 'rpe_corr_lookup')
                return keep[2]
            launch.keep, launch.op = keep, (_lib.OP_CORR_LOOKUP, a)
            return launch
        check(lib().rpe_corr_lookup(ptr(self.buf), ptr(co), self.b, self.h8, self.w8, self.levels, self.radius, ptr(out),
                                    stream_ptr()), 'rpe_corr_lookup')
        return out

    def lookup_conv1x1(self, coords, packed, out, out2=None, relu=True, prepare=False):
        """rpe_corr_lookup_conv1x1: act(convc1(lookup(coords))) in one kernel (``packed`` = PackedLookupConv of convc1's weight), written to the
        channel slices ``out`` / ``out2`` (b, 256, h8, w8); bit-identical to lookup() + conv1x1.  ``prepare=True``: a launcher with ``.op``."""
        co = _dev(coords, torch.float32, 'coords')
        if co is not coords or tuple(co.shape) != (self.b, 2, self.h8, self.w8):
            raise _lib.RpeError('corr lookup_conv1x1: coords must be a contiguous (b,2,h8,w8) tensor')
        if not PackedLookupConv.supported(self.levels, self.radius, self.w8):
            raise _lib.RpeError('corr lookup_conv1x1: needs 4 levels, radius 4 and w8 % 8 == 0 (use lookup + conv1x1)')
        sl = []
        for name, t in (('out', out), ('out2', out2)):
            if t is None:
                sl += [None, 0]
                continue
            if tuple(t.shape) != (self.b, packed.cout, self.h8, self.w8):
                raise _lib.RpeError(f'corr lookup_conv1x1: {name} must be a ({self.b},{packed.cout},{self.h8},{self.w8}) channel slice')
            pp, bs = _chan_slice(t, name)
            sl += [pp.value, bs]
        a = _lib.LookupConv1x1Args(self.buf.data_ptr(), co.data_ptr(), self.b, self.h8, self.w8, self.levels, self.radius, packed.packed.data_ptr(),
                                   packed.bias.data_ptr() if packed.bias is not None else None, int(bool(relu)), sl[0], sl[1], sl[2], sl[3])
        fn, keep = lib().rpe_corr_lookup_conv1x1, (self, co, packed, out, out2)

        def launch():
            if _REC is not None:
                _REC.log(_lib.OP_LOOKUP_CONV1X1, a, keep)
            check(fn(a.pyramid, a.coords, a.b, a.h8, a.w8, a.levels, a.radius, a.packed, a.bias, a.relu, a.out, a.out_batch_stride, a.out2,
                     a.out2_batch_stride, stream_ptr()), 'rpe_corr_lookup_conv1x1')
            return keep[3]
        launch.keep, launch.op = keep, (_lib.OP_LOOKUP_CONV1X1, a)
        if prepare:
            return launch
        return launch()

    def taps(self, coords):
        co = _dev(coords, torch.float32, 'coords')
        win = 2 * self.radius + 1
        x0 = torch.empty(self.b, self.levels, win, self.h8 * self.w8, dtype=torch.int32, device=co.device)
        y0 = torch.empty_like(x0)
        check(lib().rpe_corr_lookup_taps(ptr(co), self.b, self.h8, self.w8, self.levels, ptr(x0), ptr(y0), stream_ptr()),
              'rpe_corr_lookup_taps')
        return x0, y0

    def rounds(self, coords):
        """Diagnostic (rpe_corr_lookup_rounds): staging rounds and requested 128-B lines per (batch item, level, group)."""
        co = _dev(coords, torch.float32, 'coords')
        ng = self.h8 * ((self.w8 + 7) // 8)
        rounds = torch.zeros(self.b, self.levels, ng, dtype=torch.int32, device=co.device)
        lines = torch.zeros_like(rounds)
        check(lib().rpe_corr_lookup_rounds(ptr(co), self.b, self.h8, self.w8, self.levels, ptr(rounds), ptr(lines), stream_ptr()),
              'rpe_corr_lookup_rounds')
        return rounds, lines

    def export_level(self, level):
        h, w = self.h8 >> level, self.w8 >> level
        dense = torch.empty(self.b * self.h8 * self.w8, h, w, dtype=torch.float32, device=self.buf.device)
        check(lib().rpe_corr_export_level(ptr(self.buf), self.b, self.h8, self.w8, self.levels, level, ptr(dense),
                                          stream_ptr()), 'rpe_corr_export_level')
        return dense


class PackedLookupConv:
    """convc1's (256, 324, 1, 1) weight in rpe_corr_lookup_conv1x1's layout (a lane's fragments of a 16-channel step as one 64-byte piece)."""

    def __init__(self, weight, bias=None):
        w = _nchw(weight.detach().contiguous(), 'weight')
        self.cout, self.cin, self.kh, self.kw = w.shape
        n = lib().rpe_corr_lookup_conv1x1_packed_floats(self.cout, self.cin) if (self.kh, self.kw) == (1, 1) else 0
        if n == 0:
            raise _lib.RpeError('PackedLookupConv: needs the (256, 324, 1, 1) weight of BasicMotionEncoder.convc1')
        self.packed = torch.empty(n, dtype=torch.float32, device=w.device)
        check(lib().rpe_corr_lookup_conv1x1_pack(ptr(w), self.cout, self.cin, ptr(self.packed), stream_ptr()), 'rpe_corr_lookup_conv1x1_pack')
        self.bias = None if bias is None else _nchw(bias.detach().contiguous(), 'bias')

    @staticmethod
    def supported(levels, radius, w8):
        return levels == 4 and radius == 4 and w8 % 8 == 0


# ------------------------------------------------------------------------------------------------- RAFT update
def _nchw(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise _lib.RpeError(f'{name}: expected a contiguous float32 NCHW tensor on the GPU')
    _on_current_device(t, name)
    return t


def gru_gates_zr(zr_pre, h_buf, c, z_out, rh_buf, bias=None, add=None):
    """z_out = sigmoid(zr_pre[:, :c] + add[:, :c] + bias[:c]); rh_buf[:, :c] = sigmoid(zr_pre[:, c:] + ...) * h_buf[:, :c]."""
    _nchw(zr_pre, 'zr_pre'); _nchw(h_buf, 'h_buf'); _nchw(z_out, 'z_out'); _nchw(rh_buf, 'rh_buf')
    if add is not None and _nchw(add, 'add').shape != zr_pre.shape:
        raise _lib.RpeError('gru_gates_zr: add must have the shape of zr_pre')
    b, c2, hh, ww = zr_pre.shape
    check(lib().rpe_gru_gates_zr(ptr(zr_pre), ptr(bias), ptr(add), ptr(h_buf), h_buf.shape[1], b, c, hh * ww, ptr(z_out),
                                 ptr(rh_buf), rh_buf.shape[1], stream_ptr()), 'rpe_gru_gates_zr')


def gru_gates_h(z, q_pre, h_buf, c, h_out, bias=None, add=None):
    """h_out[:, :c] = (1 - z) * h_buf[:, :c] + z * tanh(q_pre + add + bias)."""
    _nchw(z, 'z'); _nchw(q_pre, 'q_pre'); _nchw(h_buf, 'h_buf'); _nchw(h_out, 'h_out')
    if add is not None and _nchw(add, 'add').shape != q_pre.shape:
        raise _lib.RpeError('gru_gates_h: add must have the shape of q_pre')
    b, _, hh, ww = q_pre.shape
    check(lib().rpe_gru_gates_h(ptr(z), ptr(q_pre), ptr(bias), ptr(add), ptr(h_buf), h_buf.shape[1], b, c, hh * ww, ptr(h_out),
                                h_out.shape[1], stream_ptr()), 'rpe_gru_gates_h')


def bias_act(x, bias, relu=True, out=None, out_offset=0, out2=None, out2_offset=0):
    """act(x + bias[c]) -> channels [out_offset, out_offset + c) of ``out`` (default: in place on x) and optionally
    the same into ``out2``."""
    _nchw(x, 'x')
    b, c, hh, ww = x.shape
    out = x if out is None else _nchw(out, 'out')
    if out2 is not None:
        _nchw(out2, 'out2')
    check(lib().rpe_bias_act(ptr(x), ptr(bias), b, c, hh * ww, int(bool(relu)), ptr(out), out.shape[1], out_offset,
                             ptr(out2), out2.shape[1] if out2 is not None else 0, out2_offset, stream_ptr()), 'rpe_bias_act')
    return out


def instnorm_act(x, bias, eps=1e-5, relu=True, residual=None, out=None):
    """relu?(InstanceNorm(x + bias)), then optionally relu(residual + .) -- one pass per plane."""
    _nchw(x, 'x')
    if residual is not None:
        _nchw(residual, 'residual')
    b, c, hh, ww = x.shape
    out = x if out is None else _nchw(out, 'out')
    check(lib().rpe_instnorm_act(ptr(x), ptr(bias), b, c, hh * ww, float(eps), int(bool(relu)), ptr(residual), ptr(out),
                                 stream_ptr()), 'rpe_instnorm_act')
    return out


def affine_act(x, scale, shift, relu=True, residual=None, out=None):
    """relu?(x * scale[c] + shift[c]), then optionally relu(residual + .)."""
    _nchw(x, 'x')
    if residual is not None:
        _nchw(residual, 'residual')
    b, c, hh, ww = x.shape
    out = x if out is None else _nchw(out, 'out')
    check(lib().rpe_affine_act(ptr(x), ptr(scale), ptr(shift), b, c, hh * ww, int(bool(relu)), ptr(residual), ptr(out),
                               stream_ptr()), 'rpe_affine_act')
    return out


def conv3x3_to2(x, weight, bias, add=None, out=None):
    """out = conv3x3(x, weight (2,c,3,3), padding=1) + bias [+ add]; the flow head's last layer."""
    _nchw(x, 'x')
    b, c, hh, ww = x.shape
    weight = _nchw(weight.detach() if weight.requires_grad else weight, 'weight')
    if tuple(weight.shape) != (2, c, 3, 3):
        raise _lib.RpeError('conv3x3_to2: weight must be (2,c,3,3)')
    if add is not None:
        _nchw(add, 'add')
    if out is None:
        out = torch.empty(b, 2, hh, ww, dtype=torch.float32, device=x.device)
    check(lib().rpe_conv3x3_to2(ptr(x), ptr(weight), ptr(bias), b, c, hh, ww, ptr(add), ptr(out), stream_ptr()), 'rpe_conv3x3_to2')
    return out


def flow_update(x, weight, bias, coords, coords_out, flow_out=None, dst1=None, dst2=None, prepare=False):
    """rpe_conv3x3_to2_flow: coords_out = conv3x3(x; weight (2,c,3,3)) + bias + coords, and flow = coords_out - pixel grid written to
    ``flow_out`` (b,2,h,w) and into the two-channel slices ``dst1`` / ``dst2`` (e.g. hx[:, 254:256]).  ``prepare=True`` returns a launcher."""
    _nchw(x, 'x')
    b, c, hh, ww = x.shape
    weight = _nchw(weight.detach() if weight.requires_grad else weight, 'weight')
    if tuple(weight.shape) != (2, c, 3, 3):
        raise _lib.RpeError('flow_update: weight must be (2,c,3,3)')
    for name, t in (('coords', coords), ('coords_out', coords_out), ('flow_out', flow_out)):
        if t is not None and tuple(_nchw(t, name).shape) != (b, 2, hh, ww):
            raise _lib.RpeError(f'flow_update: {name} must be ({b},2,{hh},{ww})')
    sl = []
    for name, t in (('dst1', dst1), ('dst2', dst2)):
        if t is None:
            sl += [None, 0]
            continue
        if tuple(t.shape) != (b, 2, hh, ww):
            raise _lib.RpeError(f'flow_update: {name} must be a ({b},2,{hh},{ww}) channel slice')
        sl += list(_chan_slice(t, name))
    args = (ptr(x), ptr(weight), ptr(bias), b, c, hh, ww, ptr(coords), ptr(coords_out), ptr(flow_out), sl[0], sl[1], sl[2], sl[3])
    fn = lib().rpe_conv3x3_to2_flow
    if prepare:
        keep = (x, weight, bias, coords, coords_out, flow_out, dst1, dst2)
        a = _lib.FlowUpdateArgs(*[v.value if hasattr(v, 'value') else v for v in args])

        def launch():
            if _REC is not None:
                _REC.log(_lib.OP_FLOW_UPDATE, a, keep)
            st = fn(*args, stream_ptr())
            if st != 0:
                check(st, 'rpe_conv3x3_to2_flow')
            return keep[4]
        launch.keep, launch.op = keep, (_lib.OP_FLOW_UPDATE, a)
        return launch
    check(fn(*args, stream_ptr()), 'rpe_conv3x3_to2_flow')
    return coords_out


def copy_planes(src, dst):
    """dst[:, :c] = src for channel slices of NCHW buffers (rpe_copy_planes)."""
    sp, sbs = _chan_slice(src, 'src')
    dp, dbs = _chan_slice(dst, 'dst')
    b, c, hh, ww = src.shape
    if tuple(dst.shape) != (b, c, hh, ww):
        raise _lib.RpeError('copy_planes: shape mismatch')
    if _REC is not None:
        _REC.log(_lib.OP_COPY_PLANES, _lib.CopyPlanesArgs(sp.value, sbs, dp.value, dbs, b, c, hh * ww), (src, dst))
    check(lib().rpe_copy_planes(sp, sbs, dp, dbs, b, c, hh * ww, stream_ptr()), 'rpe_copy_planes')
    return dst


def upsample_convex(flow, mask):
    fl, mk = _dev(flow, torch.float32, 'flow'), _dev(mask, torch.float32, 'mask')
    b, _, h8, w8 = fl.shape
    if tuple(mk.shape) != (b, 576, h8, w8):
        raise _lib.RpeError('upsample_convex: mask must be (b,576,h/8,w/8)')
    out = torch.empty(b, 2, 8 * h8, 8 * w8, dtype=torch.float32, device=fl.device)
    if _REC is not None:
        _REC.log(_lib.OP_UPSAMPLE_CONVEX, _lib.UpsampleConvexArgs(fl.data_ptr(), mk.data_ptr(), b, h8, w8, out.data_ptr()), (fl, mk, out))
    check(lib().rpe_upsample_convex(ptr(fl), ptr(mk), b, h8, w8, ptr(out), stream_ptr()), 'rpe_upsample_convex')
    return out


# ------------------------------------------------------------------------- fused update-block convolutions
CONV_LINEAR, CONV_RELU, CONV_GATE_ZR, CONV_GATE_H, CONV_TANH = 0, 1, 2, 3, 4


def _chan_slice(t, name):
    """(pointer, batch stride) of a float32 GPU tensor that is a channel slice ``buf[:, a:b]`` of a contiguous NCHW
    buffer (or such a buffer itself)."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4):
        raise _lib.RpeError(f'{name}: expected a float32 NCHW tensor on the GPU')
    _, _, hh, ww = t.shape
    if t.stride(3) != 1 or t.stride(2) != ww or t.stride(1) != hh * ww:
        raise _lib.RpeError(f'{name}: expected a channel slice of a contiguous NCHW buffer')
    _on_current_device(t, name)
    return ptr(t), t.stride(0)


class PackedConv:
    """Weights of one stride-1 'same' convolution re-laid for rpe_conv_fused (done once per weight version)."""

    def __init__(self, weight, bias=None):
        w = _nchw(weight.detach().contiguous(), 'weight')
        self.cout, self.cin, self.kh, self.kw = w.shape
        n = lib().rpe_conv_packed_floats(self.cout, self.cin, self.kh, self.kw)
        self.packed = torch.empty(n, dtype=torch.float32, device=w.device)
        check(lib().rpe_conv_pack(ptr(w), self.cout, self.cin, self.kh, self.kw, ptr(self.packed), stream_ptr()), 'rpe_conv_pack')
        self.bias = None if bias is None else _nchw(bias.detach().contiguous(), 'bias')

    @staticmethod
    def supported(weight, width):
        _, _, kh, kw = weight.shape
        return kh % 2 == 1 and kw in (1, 3, 5) and width % 4 == 0


class PackedConv1x1:
    """Weights of a 1x1 stride-1 convolution in rpe_conv1x1's layout ([co tile][16-channel step][k][128 co]); same attributes as
    PackedConv, so conv_fused(..., entry='rpe_conv1x1') builds the descriptor."""

    def __init__(self, weight, bias=None):
        w = _nchw(weight.detach().contiguous(), 'weight')
        self.cout, self.cin, self.kh, self.kw = w.shape
        if (self.kh, self.kw) != (1, 1):
            raise _lib.RpeError('PackedConv1x1: weight must be (cout, cin, 1, 1)')
        self.packed = torch.empty(lib().rpe_conv1x1_packed_floats(self.cout, self.cin), dtype=torch.float32, device=w.device)
        check(lib().rpe_conv1x1_pack(ptr(w), self.cout, self.cin, ptr(self.packed), stream_ptr()), 'rpe_conv1x1_pack')
        self.bias = None if bias is None else _nchw(bias.detach().contiguous(), 'bias')

    @staticmethod
    def supported(h, w):
        return (h * w) % 4 == 0 and h * w >= 4


class PackedConv1x1X3:
    """Weights of a 1x1 stride-1 convolution split three ways into bf16 for rpe_conv1x1_x3 (the labelled bf16x3 variant of rpe_conv1x1;
    conv1x1 dispatches on the packing).  LINEAR / RELU only."""
    x3 = True

    def __init__(self, weight, bias=None):
        w = _nchw(weight.detach().contiguous(), 'weight')
        self.cout, self.cin, self.kh, self.kw = w.shape
        if (self.kh, self.kw) != (1, 1):
            raise _lib.RpeError('PackedConv1x1X3: weight must be (cout, cin, 1, 1)')
        self.packed = torch.empty(lib().rpe_conv1x1_x3_packed_bytes(self.cout, self.cin) // 4, dtype=torch.float32, device=w.device)
        check(lib().rpe_conv1x1_x3_pack(ptr(w), self.cout, self.cin, ptr(self.packed), stream_ptr()), 'rpe_conv1x1_x3_pack')
        self.bias = None if bias is None else _nchw(bias.detach().contiguous(), 'bias')


def conv1x1(x, pc, mode, out, out2=None, prepare=False):
    """rpe_conv1x1: out = act(W x + bias) for a PackedConv1x1 (LINEAR / RELU / TANH), channel-slice destinations like conv_fused
    (a PackedConv1x1X3 runs rpe_conv1x1_x3)."""
    return conv_fused(x, pc, mode, out, out2=out2, prepare=prepare, entry='rpe_conv1x1_x3' if getattr(pc, 'x3', False) else 'rpe_conv1x1')


class Conv1x1:
    """A 1x1 layer with both packings, each made on first use: launches of at least 512 workgroups of 128 x 128 run on rpe_conv1x1
    (LDS-DMA GEMM: convc1 337 -> 260 us at batch 32), smaller ones on rpe_conv_fused, whose 64 x 64 tiles fill the chip better (batch 2:
    29 vs 32 us).  The two kernels sum the same products in the same order (bit-identical results), so the choice never shows."""

    def __init__(self, weight, bias=None):
        self._w = _nchw(weight.detach().contiguous(), 'weight')
        self._b = None if bias is None else _nchw(bias.detach().contiguous(), 'bias')
        self.cout, self.cin = self._w.shape[0], self._w.shape[1]
        self._fused = self._gemm = self._gemm_x3 = None

    @property
    def fused(self):
        if self._fused is None:
            self._fused = PackedConv(self._w, self._b)
        return self._fused

    @property
    def gemm(self):
        if self._gemm is None:
            self._gemm = PackedConv1x1(self._w, self._b)
        return self._gemm

    @property
    def gemm_x3(self):
        if self._gemm_x3 is None:
            self._gemm_x3 = PackedConv1x1X3(self._w, self._b)
        return self._gemm_x3

    def __call__(self, x, mode, out, out2=None, prepare=False, x3=False):
        b, _, hh, ww = x.shape
        # rpe_conv1x1's own preconditions (16-byte DMA pieces): plane size, base and batch stride of the input slice
        aligned = PackedConv1x1.supported(hh, ww) and x.data_ptr() % 16 == 0 and x.stride(0) % 4 == 0
        big = b * -(-(hh * ww) // 128) * -(-self.cout // 128) >= 512 and aligned
        if aligned and x3 and mode in (CONV_LINEAR, CONV_RELU):     # the labelled bf16x3 variant (raft.CONV_BF16X3): at EVERY launch size, so that a
            #                                                            row's bits do not depend on the batch it is launched in
            return conv1x1(x, self.gemm_x3, mode, out, out2=out2, prepare=prepare)
        if big:
            return conv1x1(x, self.gemm, mode, out, out2=out2, prepare=prepare)
        return conv_fused(x, self.fused, mode, out, out2=out2, prepare=prepare)


def conv_fused(x, pc, mode, out, out2=None, add=None, hidden=None, zgate=None, gate_channels=0, scale=None, bias='packed',
               residual=None, stats=None, stride=1, pre_norm=None, prepare=False, entry='rpe_conv_fused'):
    """rpe_conv_fused: out = epilogue(conv(x; pc) * scale + add + bias).  All tensors are channel slices of NCHW buffers.
    ``bias`` defaults to the one packed with the weights; ``stats`` (from conv_stats_buffer) collects the per-tile moments
    instnorm_apply needs.  ``prepare=True`` returns a zero-argument launcher instead of launching: the GRU loop runs the same
    nine convolutions on the same buffers twelve times, and at batch 1 the Python argument checking costs more than the kernels."""
    d = _lib.ConvDesc()
    b, cin, hh, ww = x.shape
    if cin != pc.cin:
        raise _lib.RpeError(f'conv_fused: input has {cin} channels, weights expect {pc.cin}')
    d.x, d.x_batch_stride = _chan_slice(x, 'x')
    bias = pc.bias if isinstance(bias, str) else bias
    for name, t in (('bias', bias), ('scale', scale)):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != pc.cout):
            raise _lib.RpeError(f'conv_fused: {name} must be a contiguous float32 GPU vector of cout elements')
    d.packed, d.bias, d.scale = ptr(pc.packed), ptr(bias), ptr(scale)
    for name, t, want_c in (('add', add, pc.cout), ('out', out, None), ('out2', out2, None), ('hidden', hidden, None), ('zgate', zgate, None),
                            ('residual', residual, pc.cout)):
        if t is None:
            setattr(d, name, None); setattr(d, name + '_batch_stride', 0)
            continue
        if t.shape[0] != b or tuple(t.shape[2:]) != (hh // stride, ww // stride) or (want_c is not None and t.shape[1] != want_c):
            raise _lib.RpeError(f'conv_fused: {name} has shape {tuple(t.shape)}')
        p, s = _chan_slice(t, name)
        setattr(d, name, p); setattr(d, name + '_batch_stride', s)
    need = {CONV_LINEAR: pc.cout, CONV_RELU: pc.cout, CONV_GATE_ZR: gate_channels, CONV_GATE_H: pc.cout, CONV_TANH: pc.cout}[mode]
    if out.shape[1] < need or (mode == CONV_GATE_ZR and (out2 is None or out2.shape[1] < gate_channels or hidden.shape[1] < gate_channels)):
        raise _lib.RpeError('conv_fused: destination slice has too few channels')
    if mode == CONV_GATE_H and (hidden is None or zgate is None or hidden.shape[1] < pc.cout or zgate.shape[1] < pc.cout):
        raise _lib.RpeError('conv_fused: GATE_H needs hidden and zgate with cout channels')
    if mode in (CONV_LINEAR, CONV_RELU, CONV_TANH) and out2 is not None and out2.shape[1] < pc.cout:
        raise _lib.RpeError('conv_fused: out2 slice has too few channels')
    if stats is not None:
        tiles = lib().rpe_conv_stats_tiles(pc.cout, hh, ww, stride)
        if not (stats.is_cuda and stats.dtype == torch.float32 and stats.is_contiguous() and tuple(stats.shape) == (b, pc.cout, tiles, 3)):
            raise _lib.RpeError(f'conv_fused: stats must be a contiguous float32 ({b},{pc.cout},{tiles},3) GPU tensor (conv_stats_buffer)')
        d.stats_tiles = tiles
    d.stats = ptr(stats)
    if pre_norm is not None and not (pre_norm.is_cuda and pre_norm.dtype == torch.float32 and pre_norm.is_contiguous()
                                     and tuple(pre_norm.shape) == (b, cin, 2)):
        raise _lib.RpeError(f'conv_fused: pre_norm must be a contiguous float32 ({b},{cin},2) GPU tensor')
    d.pre_norm = ptr(pre_norm)
    d.b, d.cin, d.cout, d.h, d.w, d.kh, d.kw, d.mode, d.gate_channels = b, cin, pc.cout, hh, ww, pc.kh, pc.kw, mode, gate_channels
    d.stride = stride
    import ctypes
    if _REC is not None and not prepare:
        _REC.log(_lib.OP_OF_ENTRY[entry], d, (x, pc, out, out2, add, hidden, zgate, scale, bias, residual, stats, pre_norm))
    if prepare:                                    # the checked descriptor, to be launched again and again on the same buffers
        fn, ref, keep = getattr(lib(), entry), ctypes.byref(d), (d, x, pc, out, out2, add, hidden, zgate, scale, bias, residual, stats, pre_norm)

        def launch():
            if _REC is not None:
                _REC.log(_lib.OP_OF_ENTRY[entry], d, keep)
            st = fn(ref, stream_ptr())
            if st != 0:
                check(st, entry)
            return keep[3]
        launch.keep, launch.op = keep, (_lib.OP_OF_ENTRY[entry], d)
        return launch
    check(getattr(lib(), entry)(ctypes.byref(d), stream_ptr()), entry)
    return out


def conv_direct(x, weight, bias=None, stride=1, padding=0, relu=False, out=None):
    """rpe_conv_direct: torch.nn.functional.conv2d(x, weight, bias, stride, padding) [+ ReLU] for ANY map size -- the route of the shapes
    the tuned kernels refuse (odd maps, rows that are not whole 16-byte quads).  x / out may be channel slices of NCHW buffers."""
    xp, xbs = _chan_slice(x, 'x')
    b, cin, hh, ww = x.shape
    w = _nchw(weight.detach().contiguous(), 'weight')
    cout, wcin, kh, kw = w.shape
    if wcin != cin:
        raise _lib.RpeError(f'conv_direct: input has {cin} channels, weight expects {wcin}')
    st = stride if isinstance(stride, int) else stride[0]
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    ho, wo = (hh + 2 * ph - kh) // st + 1, (ww + 2 * pw - kw) // st + 1
    if bias is not None:
        bias = bias.detach()
        if not (bias.is_cuda and bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == cout):
            raise _lib.RpeError('conv_direct: bias must be a contiguous float32 GPU vector of cout elements')
    if out is None:
        out = torch.empty(b, cout, ho, wo, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (b, cout, ho, wo):
        raise _lib.RpeError(f'conv_direct: out must be ({b},{cout},{ho},{wo})')
    op, obs = _chan_slice(out, 'out')
    check(lib().rpe_conv_direct(xp, xbs, ptr(w), ptr(bias), b, cin, cout, hh, ww, kh, kw, st, ph, pw, int(bool(relu)), op, obs, stream_ptr()),
          'rpe_conv_direct')
    return out


class PackedWino1d:
    """Weights of a 1x5 / 5x1 stride-1 convolution transformed for rpe_conv_wino1d (U = G g along the taps, once per weight version)."""

    def __init__(self, weight, bias=None):
        w = _nchw(weight.detach().contiguous(), 'weight')
        self.cout, self.cin, self.kh, self.kw = w.shape
        n = lib().rpe_conv_wino1d_packed_floats(self.cout, self.cin) if (self.kh, self.kw) in ((1, 5), (5, 1)) else 0
        if n == 0:
            raise _lib.RpeError('PackedWino1d: needs a (cout, cin % 4 == 0, 1, 5) or (.., 5, 1) weight')
        self.packed = torch.empty(n, dtype=torch.float32, device=w.device)
        check(lib().rpe_conv_wino1d_pack(ptr(w), self.cout, self.cin, ptr(self.packed), stream_ptr()), 'rpe_conv_wino1d_pack')
        self.bias = None if bias is None else _nchw(bias.detach().contiguous(), 'bias')

    @staticmethod
    def supported(weight, ww):
        return tuple(weight.shape[2:]) in ((1, 5), (5, 1)) and weight.shape[1] % 4 == 0 and ww % 4 == 0


def conv_wino1d(x, pw, mode, out, **kw):
    """rpe_conv_wino1d: conv_fused's operation (all four epilogue modes incl. the GRU gates) for 1x5 / 5x1 stride-1 convolutions
    by Winograd F(4,5) along the filter axis: 2.5x fewer matrix FLOPs.  Same keyword arguments as conv_fused (no scale /
    residual / stats / pre_norm)."""
    return conv_fused(x, pw, mode, out, entry='rpe_conv_wino1d_x3' if getattr(pw, 'x3', False) else 'rpe_conv_wino1d', **kw)


class PackedWino1dX3:
    """Weights of a 1x5 / 5x1 stride-1 convolution transformed and split three ways into bf16 for rpe_conv_wino1d_x3 (the labelled
    bf16x3 variant of rpe_conv_wino1d; conv_wino1d dispatches on the packing)."""
    x3 = True

    def __init__(self, weight, bias=None):
        w = _nchw(weight.detach().contiguous(), 'weight')
        self.cout, self.cin, self.kh, self.kw = w.shape
        n = lib().rpe_conv_wino1d_x3_packed_bytes(self.cout, self.cin) if (self.kh, self.kw) in ((1, 5), (5, 1)) else 0
        if n == 0:
            raise _lib.RpeError('PackedWino1dX3: needs a (cout, cin % 16 == 0, 1, 5) or (.., 5, 1) weight')
        self.packed = torch.empty(n // 4, dtype=torch.float32, device=w.device)
        check(lib().rpe_conv_wino1d_x3_pack(ptr(w), self.cout, self.cin, ptr(self.packed), stream_ptr()), 'rpe_conv_wino1d_x3_pack')
        self.bias = None if bias is None else _nchw(bias.detach().contiguous(), 'bias')

    @staticmethod
    def supported(weight, ww):
        return tuple(weight.shape[2:]) in ((1, 5), (5, 1)) and weight.shape[1] % 16 == 0 and ww % 4 == 0


class PackedWino:
    """Weights of a 3x3 stride-1 convolution transformed for rpe_conv_wino (U = G g G^T, once per weight version)."""

    def __init__(self, weight, bias=None):
        w = _nchw(weight.detach().contiguous(), 'weight')
        self.cout, self.cin, self.kh, self.kw = w.shape
        n = lib().rpe_conv_wino_packed_floats(self.cout, self.cin) if (self.kh, self.kw) == (3, 3) else 0
        if n == 0:
            raise _lib.RpeError('PackedWino: needs a (cout, cin % 4 == 0, 3, 3) weight')
        self.packed = torch.empty(n, dtype=torch.float32, device=w.device)
        check(lib().rpe_conv_wino_pack(ptr(w), self.cout, self.cin, ptr(self.packed), stream_ptr()), 'rpe_conv_wino_pack')
        self.bias = None if bias is None else _nchw(bias.detach().contiguous(), 'bias')

    @staticmethod
    def supported(weight, hh, ww):
        return tuple(weight.shape[2:]) == (3, 3) and weight.shape[1] % 4 == 0 and hh % 2 == 0 and ww % 2 == 0


class PackedWinoX3:
    """Weights of a 3x3 stride-1 convolution for rpe_conv_wino_x3, the LABELLED bf16x3 variant of rpe_conv_wino: U = G g G^T in f32,
    then the exact three-way bf16 split.  conv_wino() takes either packing and runs the kernel that belongs to it."""
    x3 = True

    def __init__(self, weight, bias=None):
        w = _nchw(weight.detach().contiguous(), 'weight')
        self.cout, self.cin, self.kh, self.kw = w.shape
        n = lib().rpe_conv_wino_x3_packed_bytes(self.cout, self.cin) if (self.kh, self.kw) == (3, 3) else 0
        if n == 0:
            raise _lib.RpeError('PackedWinoX3: needs a (cout, cin % 16 == 0, 3, 3) weight')
        self.packed = torch.empty(n // 4, dtype=torch.float32, device=w.device)       # (three bf16 planes; float32 storage for the descriptor)
        check(lib().rpe_conv_wino_x3_pack(ptr(w), self.cout, self.cin, ptr(self.packed), stream_ptr()), 'rpe_conv_wino_x3_pack')
        self.bias = None if bias is None else _nchw(bias.detach().contiguous(), 'bias')

    @staticmethod
    def supported(weight, hh, ww):
        return tuple(weight.shape[2:]) == (3, 3) and weight.shape[1] % 16 == 0 and hh % 2 == 0 and ww % 4 == 0


def conv_wino(x, pw, mode, out, out2=None, scale=None, bias='packed', residual=None, stats=None, pre_norm=None, prepare=False):
    """rpe_conv_wino: out = epilogue(conv3x3(x; pw) * scale + bias) by Winograd F(2x2,3x3); tensors are channel slices of NCHW
    buffers.  ``stats`` (conv_wino_stats_buffer) / ``pre_norm`` / ``residual`` / ``scale``: the encoders' epilogues, as conv_fused."""
    import ctypes
    d = _lib.ConvDesc()
    b, cin, hh, ww = x.shape
    if cin != pw.cin:
        raise _lib.RpeError(f'conv_wino: input has {cin} channels, weights expect {pw.cin}')
    if mode not in (CONV_LINEAR, CONV_RELU):
        raise _lib.RpeError('conv_wino: LINEAR / RELU epilogues only')
    d.x, d.x_batch_stride = _chan_slice(x, 'x')
    bias = pw.bias if isinstance(bias, str) else bias
    for name, t in (('bias', bias), ('scale', scale)):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != pw.cout):
            raise _lib.RpeError(f'conv_wino: {name} must be a contiguous float32 GPU vector of cout elements')
    d.packed, d.bias, d.scale = ptr(pw.packed), ptr(bias), ptr(scale)
    for name, t in (('out', out), ('out2', out2), ('residual', residual)):
        if t is None:
            setattr(d, name, None); setattr(d, name + '_batch_stride', 0)
            continue
        if t.shape[0] != b or tuple(t.shape[2:]) != (hh, ww) or t.shape[1] < pw.cout:
            raise _lib.RpeError(f'conv_wino: {name} has shape {tuple(t.shape)}')
        p, s = _chan_slice(t, name)
        setattr(d, name, p); setattr(d, name + '_batch_stride', s)
    if stats is not None:
        tiles = lib().rpe_conv_wino_stats_tiles(hh, ww)
        if not isinstance(stats, TileMajorStats):
            raise _lib.RpeError('conv_wino: stats must come from conv_wino_stats_buffer (tile-major records)')
        stats = stats.tensor
        if not (stats.is_cuda and stats.dtype == torch.float32 and stats.is_contiguous() and tuple(stats.shape) == (b, tiles, pw.cout, 3)):
            raise _lib.RpeError(f'conv_wino: stats must be a contiguous float32 ({b},{tiles},{pw.cout},3) GPU tensor (conv_wino_stats_buffer)')
    if pre_norm is not None and not (pre_norm.is_cuda and pre_norm.dtype == torch.float32 and pre_norm.is_contiguous()
                                     and tuple(pre_norm.shape) == (b, cin, 2)):
        raise _lib.RpeError(f'conv_wino: pre_norm must be a contiguous float32 ({b},{cin},2) GPU tensor')
    d.stats, d.pre_norm = ptr(stats), ptr(pre_norm)
    d.b, d.cin, d.cout, d.h, d.w, d.kh, d.kw, d.mode, d.stride = b, cin, pw.cout, hh, ww, 3, 3, mode, 1
    if _REC is not None and not prepare:
        _REC.log(_lib.OP_CONV_WINO_X3 if getattr(pw, 'x3', False) else _lib.OP_CONV_WINO, d, (x, pw, out, out2, scale, bias, residual, stats, pre_norm))
    if prepare:
        fn, ref, keep = (lib().rpe_conv_wino_x3 if getattr(pw, 'x3', False) else lib().rpe_conv_wino), ctypes.byref(d), (d, x, pw, out, out2, scale, bias, residual, stats, pre_norm)

        def launch():
            if _REC is not None:
                _REC.log(_lib.OP_CONV_WINO_X3 if getattr(pw, 'x3', False) else _lib.OP_CONV_WINO, d, keep)
            st = fn(ref, stream_ptr())
            if st != 0:
                check(st, 'rpe_conv_wino')
            return keep[3]
        launch.keep, launch.op = keep, (_lib.OP_CONV_WINO_X3 if getattr(pw, 'x3', False) else _lib.OP_CONV_WINO, d)
        return launch
    if getattr(pw, 'x3', False):
        check(lib().rpe_conv_wino_x3(ctypes.byref(d), stream_ptr()), 'rpe_conv_wino_x3')
    else:
        check(lib().rpe_conv_wino(ctypes.byref(d), stream_ptr()), 'rpe_conv_wino')
    return out


class TileMajorStats:
    """The (b, tiles, cout, 3) per-tile (count, mean, M2) records of rpe_conv_wino.  The layout is part of the TYPE, not an
    attribute a view or clone could drop: instnorm_finalize / instnorm_apply take either this (tile-major) or a plain
    (b, cout, tiles, 3) tensor (rpe_conv_fused's and rpe_stem_conv's channel-major records) and validate the shape for each."""

    def __init__(self, tensor):
        self.tensor = tensor

    @property
    def shape(self):
        return self.tensor.shape

    def cpu(self):
        return self.tensor.cpu()


def conv_wino_stats_buffer(b, cout, hh, ww, device):
    """Per-tile (count, mean, M2) records rpe_conv_wino fills when ``stats`` is given: TILE-MAJOR (b, tiles, cout, 3)."""
    return TileMajorStats(torch.empty(b, lib().rpe_conv_wino_stats_tiles(hh, ww), cout, 3, dtype=torch.float32, device=device))


def _stats_layout(stats, b, c, who):
    """(tensor, signed tile count for the C ABI: negative = tile-major) after validating the records' shape against (b, c)."""
    tile_major = isinstance(stats, TileMajorStats)
    t = stats.tensor if tile_major else stats
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 4 and t.shape[3] == 3):
        raise _lib.RpeError(f'{who}: stats must be a contiguous float32 4-D GPU tensor of (count, mean, M2) records')
    want = (b, t.shape[1], c) if tile_major else (b, c, t.shape[2])
    if tuple(t.shape[:3]) != want or t.shape[1 if tile_major else 2] < 1:
        raise _lib.RpeError(f'{who}: stats must be the (b,c,tiles,3) buffer of conv_fused / stem_conv or the TileMajorStats (b,tiles,c,3) of conv_wino; '
                            f'got {tuple(t.shape)} for b={b}, c={c}')
    return t, (-t.shape[1] if tile_major else t.shape[2])


def conv_stats_buffer(b, cout, hh, ww, device, stride=1):
    """Per-tile (count, mean, M2) records rpe_conv_fused fills when ``stats`` is given: (b, cout, tiles, 3); hh, ww = input map."""
    return torch.empty(b, cout, lib().rpe_conv_stats_tiles(cout, hh, ww, stride), 3, dtype=torch.float32, device=device)


def instnorm_finalize(stats, hw, eps=1e-5, channels=None):
    """(b,c,2) = (mean, 1/std) per plane from conv_fused's partial sums: the ``pre_norm`` argument of the next conv_fused.
    ``channels`` (optional) is checked against the records' channel axis."""
    b = stats.shape[0]
    c = stats.shape[2] if isinstance(stats, TileMajorStats) else stats.shape[1]
    if channels is not None and channels != c:
        raise _lib.RpeError(f'instnorm_finalize: records hold {c} channels, expected {channels}')
    t, tiles = _stats_layout(stats, b, c, 'instnorm_finalize')
    mi = torch.empty(b, c, 2, dtype=torch.float32, device=t.device)
    if _REC is not None:
        _REC.log(_lib.OP_INSTNORM_FINALIZE, _lib.InstnormFinalizeArgs(t.data_ptr(), tiles, b, c, hw, float(eps), mi.data_ptr()), (t, mi))
    check(lib().rpe_instnorm_finalize(ptr(t), tiles, b, c, hw, float(eps), ptr(mi), stream_ptr()), 'rpe_instnorm_finalize')
    return mi


def instnorm_apply(x, stats, eps=1e-5, relu=True, residual=None, out=None, residual_norm=None, residual_relu=True):
    """Instance norm of x (b,c,h,w) from the partial sums of conv_fused(..., stats=stats): one read + one write pass.
    ``stats`` may also be the (b,c,2) result of instnorm_finalize on those sums (``eps`` is then already in it): the pass is then a
    pure stream with several workgroups per plane, the faster form behind large maps (the records are merged once, not per workgroup).
    ``residual_norm`` (b,c,2) from instnorm_finalize: the residual is a RAW convolution output, normalised (+ ReLU'd unless
    ``residual_relu=False``: a stride-2 block's shortcut has none) on the fly."""
    _nchw(x, 'x')
    b, c, hh, ww = x.shape
    if isinstance(stats, torch.Tensor) and stats.dim() == 3:              # (mean, 1/std) pairs of instnorm_finalize: a pure streaming pass
        if not (stats.is_cuda and stats.dtype == torch.float32 and stats.is_contiguous() and tuple(stats.shape) == (b, c, 2)):
            raise _lib.RpeError(f'instnorm_apply: precomputed moments must be a contiguous float32 ({b},{c},2) GPU tensor (instnorm_finalize)')
        t, tiles = stats, 0
    else:
        t, tiles = _stats_layout(stats, b, c, 'instnorm_apply')
    if residual is not None and _nchw(residual, 'residual').shape != x.shape:
        raise _lib.RpeError('instnorm_apply: residual must have the shape of x')
    if residual_norm is not None and (residual is None or not (residual_norm.is_cuda and residual_norm.dtype == torch.float32
                                                               and residual_norm.is_contiguous() and tuple(residual_norm.shape) == (b, c, 2))):
        raise _lib.RpeError(f'instnorm_apply: residual_norm must be a contiguous float32 ({b},{c},2) GPU tensor next to a residual')
    out = x if out is None else _nchw(out, 'out')
    flags = int(bool(relu)) | (0 if residual_relu or residual_norm is None else 2)
    if _REC is not None:
        dp = lambda v: v.data_ptr() if v is not None else None
        _REC.log(_lib.OP_INSTNORM_APPLY, _lib.InstnormApplyArgs(x.data_ptr(), t.data_ptr(), tiles, b, c, hh * ww, float(eps), flags, dp(residual),
                                                                dp(residual_norm), out.data_ptr()), (x, t, residual, residual_norm, out))
    check(lib().rpe_instnorm_apply_ex(ptr(x), ptr(t), tiles, b, c, hh * ww, float(eps), flags, ptr(residual), ptr(residual_norm),
                                      ptr(out), stream_ptr()), 'rpe_instnorm_apply_ex')
    return out


class PackedStem:
    """A (cout, cin, 7, 7) weight (cin 3: encoder stem, stride 2; cin 2: convf1, stride 1) in rpe_stem_conv's layout."""

    def __init__(self, weight):
        w = _nchw(weight.detach().contiguous(), 'weight')
        self.cout, self.cin = w.shape[0], w.shape[1]
        if tuple(w.shape[2:]) != (7, 7) or self.cin not in (2, 3) or self.cout % 64:
            raise _lib.RpeError('PackedStem: weight must be (64k, 2|3, 7, 7)')
        self.stride = 2 if self.cin == 3 else 1
        self.packed = torch.empty(lib().rpe_stem_packed_floats(self.cout, self.cin), dtype=torch.float32, device=w.device)
        check(lib().rpe_stem_pack(ptr(w), self.cout, self.cin, ptr(self.packed), stream_ptr()), 'rpe_stem_pack')


def stem_conv(image, ps, bias=None, scale=None, relu=True, stats=False, div=255.0, mul=2.0, sub=1.0, out=None, prepare=False):
    """conv7x7(mul * (image / div) - sub) * scale + bias [ReLU]; stride and channel counts come from ``ps``.
    Returns out, or (out, stats).  ``prepare=True`` (needs ``out``; stats = a caller-owned buffer or False): a launcher with ``.op``."""
    _nchw(image, 'image')
    b, c, hh, ww = image.shape
    if c != ps.cin:
        raise _lib.RpeError(f'stem_conv: image must have {ps.cin} channels')
    st_ = ps.stride
    if out is None:
        out = torch.empty(b, ps.cout, hh // st_, ww // st_, dtype=torch.float32, device=image.device)
    elif tuple(_nchw(out, 'out').shape) != (b, ps.cout, hh // st_, ww // st_):
        raise _lib.RpeError('stem_conv: out has the wrong shape')
    if isinstance(stats, torch.Tensor):             # a caller-owned (batch slice of a) statistics buffer
        st = stats
        if not (st.is_cuda and st.dtype == torch.float32 and st.is_contiguous() and tuple(st.shape) == (b, ps.cout, lib().rpe_stem_tiles(hh, ww, st_), 3)):
            raise _lib.RpeError('stem_conv: stats buffer has the wrong shape')
        stats = True
    else:
        st = torch.empty(b, ps.cout, lib().rpe_stem_tiles(hh, ww, st_), 3, dtype=torch.float32, device=image.device) if stats else None
    if prepare or _REC is not None:
        dp = lambda t: t.data_ptr() if t is not None else None
        a = _lib.StemConvArgs(dp(image), b, c, hh, ww, st_, float(div), float(mul), float(sub), dp(ps.packed), ps.cout, dp(bias), dp(scale), int(bool(relu)),
                              dp(out), dp(st))
        fn, keep = lib().rpe_stem_conv, (image, ps, bias, scale, out, st)
        if not prepare:
            _REC.log(_lib.OP_STEM_CONV, a, keep)

    if prepare:
        def launch():
            if _REC is not None:
                _REC.log(_lib.OP_STEM_CONV, a, keep)
            check(fn(a.image, a.b, a.cin, a.h, a.w, a.stride, a.div, a.mul, a.sub, a.packed, a.cout, a.bias, a.scale, a.relu, a.out, a.stats, stream_ptr()),
                  'rpe_stem_conv')
            return keep[4]
        launch.keep, launch.op = keep, (_lib.OP_STEM_CONV, a)
        return launch
    check(lib().rpe_stem_conv(ptr(image), b, c, hh, ww, st_, float(div), float(mul), float(sub), ptr(ps.packed), ps.cout, ptr(bias), ptr(scale),
                              int(bool(relu)), ptr(out), ptr(st), stream_ptr()), 'rpe_stem_conv')
    return (out, st) if stats else out


# ------------------------------------------------------------------------------------------------- weight heads
def unet_heads(inp1, inp2, hidden, context, params2d, params3d, out_size):
    """Both TinyUNet weight heads + resize + sigmoid (rpe_unet_heads).  hidden / context may be channel slices of (b,128,..)."""
    i1, i2 = _nchw(inp1, 'inp1'), _nchw(inp2, 'inp2')
    b, _, h8, w8 = i1.shape
    hp, hbs = _chan_slice(hidden, 'hidden')
    cp, cbs = _chan_slice(context, 'context')
    if tuple(i1.shape) != (b, 8, h8, w8) or tuple(i2.shape) != (b, 8, h8, w8) or tuple(hidden.shape) != (b, 128, h8, w8) or \
            tuple(context.shape) != (b, 128, h8, w8):
        raise _lib.RpeError('unet_heads: inp1, inp2 must be (b,8,h/8,w/8), hidden and context (b,128,h/8,w/8)')
    for name, t, cin in (('params2d', params2d, 264), ('params3d', params3d, 272)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == lib().rpe_unet_params_floats(cin)):
            raise _lib.RpeError(f'unet_heads: {name} must be the packed float32 parameter blob of TinyUNet({cin})')
    nws = lib().rpe_unet_workspace_bytes(b, h8, w8)
    if nws == 0:
        raise _lib.RpeError('unet_heads: the 1/8 grid is too small for the valid convolutions (needs >= 44x44)')
    H, W = out_size
    ws = torch.empty(nws, dtype=torch.uint8, device=i1.device)
    o2, o3 = (torch.empty(b, 1, H, W, dtype=torch.float32, device=i1.device) for _ in range(2))
    check(lib().rpe_unet_heads(ptr(i1), ptr(i2), hp, cp, hbs, cbs, ptr(params2d), ptr(params3d), b, h8, w8, H, W, ptr(o2), ptr(o3), ptr(ws),
                               stream_ptr()), 'rpe_unet_heads')
    return o2, o3
