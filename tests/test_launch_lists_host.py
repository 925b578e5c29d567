"""CPU: the host logic of the prepared launch lists (ops.OpList / ops.Recorder) without a GPU -- argument blocks are plain ctypes structs
and ``data_ptr`` works on CPU tensors, so logging, binding by address range and pointer rewriting can be checked here; nothing is
launched (the library is loaded for its symbols only)."""
import ctypes
import os

import pytest
import torch


def test_recorder_binds_by_address_range_and_rewrites_offsets(rpe):
    from rpe_amd import _lib, ops
    rec = ops.Recorder.__new__(ops.Recorder)                 # (no `with`: entering would swap in the launch counter and expect launches)
    ops.OpList.__init__(rec)
    rec._structs, rec._bound, rec._claimed, rec.complete = [], {}, set(), True
    x = torch.zeros(2, 8, 4, 4)                               # an "input": two ops read slices of it
    y = torch.zeros(2, 4, 4, 4)                               # an "output"
    w = torch.zeros(64)                                       # an intermediate the list owns: never rebound
    el = x.element_size()
    a = _lib.CopyPlanesArgs(x.data_ptr(), 128, w.data_ptr(), 64, 2, 4, 16)
    b = _lib.CopyPlanesArgs(x[:, 4:].data_ptr(), 128, y.data_ptr(), 64, 2, 4, 16)                   # channel slice: base + 4 planes
    c = _lib.InstnormFinalizeArgs(w.data_ptr(), 1, 2, 4, 16, 1e-5, y[1:].data_ptr())                  # batch slice of the output
    for kind, st in ((_lib.OP_COPY_PLANES, a), (_lib.OP_COPY_PLANES, b), (_lib.OP_INSTNORM_FINALIZE, c)):
        rec.log(kind, st, (x, y, w))
    assert len(rec) == 3 and rec.bind('x', x) == 2 and rec.bind('y', y) == 2
    assert rec.bind('x_again', x) == 0                        # a pointer is claimed once
    x2, y2 = torch.zeros(2, 8, 4, 4), torch.zeros(2, 4, 4, 4)
    rec.patch({'x': x2, 'y': y2})
    assert a.src == x2.data_ptr() and b.src == x2.data_ptr() + 4 * 16 * el and b.dst == y2.data_ptr() and c.mean_inv == y2.data_ptr() + 64 * el
    assert a.dst == w.data_ptr() and c.partials == w.data_ptr()                                      # the workspace stays where it is
    with pytest.raises(rpe.RpeError):
        rec.patch({'x': torch.zeros(2, 8, 4, 5)})             # another shape
    with pytest.raises(rpe.RpeError):
        rec.patch({'x': torch.zeros(2, 8, 4, 4).transpose(2, 3)})                                    # not contiguous
    with pytest.raises(rpe.RpeError):
        rec.patch({'x': torch.zeros(2, 8, 4, 4, dtype=torch.float64)})


def test_oplist_layout_and_marks(rpe):
    from rpe_amd import _lib, ops

    class Launcher:                                           # what ops.* returns under prepare=True, reduced to what a list reads
        def __init__(self, kind, st):
            self.op = (kind, st)
    t = torch.zeros(16)
    a = _lib.CopyPlanesArgs(t.data_ptr(), 16, t.data_ptr(), 16, 1, 1, 16)
    lst = ops.OpList(n_cells=4)
    lst.record(0, 0).wait(0, 1).add(Launcher(_lib.OP_COPY_PLANES, a), 1).record(1, 1)
    m = lst.mark()
    lst.wait(1, 0).add(Launcher(_lib.OP_COPY_PLANES, a))
    assert m == 4 and len(lst) == 6
    kinds = [k for k, _, _ in lst._items]
    assert kinds == [_lib.OP_EVENT_RECORD, _lib.OP_STREAM_WAIT, _lib.OP_COPY_PLANES, _lib.OP_EVENT_RECORD, _lib.OP_STREAM_WAIT, _lib.OP_COPY_PLANES]
    assert [s for _, s, _ in lst._items] == [0, 1, 1, 1, 0, 0]
    cell = ctypes.sizeof(ctypes.c_void_p)
    base = ctypes.addressof(lst.cells)
    assert lst._items[0][2] == base and lst._items[3][2] == base + cell and lst._items[2][2] == ctypes.addressof(a)
    # struct mirrors: the C side reads these layouts (include/rpe.h)
    assert ctypes.sizeof(_lib.Op) == 16 and _lib.Op.args.offset == 8
    assert ctypes.sizeof(_lib.CorrLookupArgs) == 48 and ctypes.sizeof(_lib.CopyPlanesArgs) == 48 and ctypes.sizeof(_lib.UpsampleConvexArgs) == 40
    assert _lib.LookupConv1x1Args.out.offset == 64 and ctypes.sizeof(_lib.LookupConv1x1Args) == 96
    # ... and csrc/oplist.hip static_asserts the C structs to the same sizes
    sizes = {'rpe_op': 16, 'rpe_conv_desc': 200, 'rpe_corr_lookup_args': 48, 'rpe_corr_build_args': 48, 'rpe_stem_conv_args': 96, 'rpe_flow_update_args': 96, 'rpe_copy_planes_args': 48, 'rpe_instnorm_finalize_args': 40, 'rpe_instnorm_apply_args': 64, 'rpe_upsample_convex_args': 40, 'rpe_lookup_conv1x1_args': 96, 'rpe_solve_opts': 32}
    mirrors = {'rpe_op': _lib.Op, 'rpe_conv_desc': _lib.ConvDesc, 'rpe_corr_lookup_args': _lib.CorrLookupArgs, 'rpe_corr_build_args': _lib.CorrBuildArgs, 'rpe_stem_conv_args': _lib.StemConvArgs, 'rpe_flow_update_args': _lib.FlowUpdateArgs, 'rpe_copy_planes_args': _lib.CopyPlanesArgs, 'rpe_instnorm_finalize_args': _lib.InstnormFinalizeArgs, 'rpe_instnorm_apply_args': _lib.InstnormApplyArgs, 'rpe_upsample_convex_args': _lib.UpsampleConvexArgs, 'rpe_lookup_conv1x1_args': _lib.LookupConv1x1Args, 'rpe_solve_opts': _lib.SolveOpts}
    assert all(ctypes.sizeof(mirrors[n]) == sizes[n] for n in sizes)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'robust-pose-estimator_amd', 'csrc', 'oplist.hip')).read()
    assert all(f'static_assert(sizeof({n}) == {v},' in src for n, v in sizes.items())
