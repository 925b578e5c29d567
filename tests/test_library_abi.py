"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/rpe.h declares.
No compute calls (there is no GPU here)."""
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'rpe.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rpe_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree(rpe):
    from rpe_amd import _lib
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_loads_and_exports_every_symbol(rpe):
    L = rpe.lib()
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert b'gfx950' in L.rpe_version()


def test_size_queries(rpe):
    L = rpe.lib()
    assert L.rpe_pose_workspace_bytes(1, 512, 640) > 0
    assert L.rpe_pose_workspace_bytes(0, 512, 640) == 0
    # level 0 alone is 5120*5120*4 B per pair
    assert L.rpe_corr_pyramid_bytes(1, 64, 80, 4) >= int(5120 * 5120 * 4 * 1.32)
    assert L.rpe_corr_pyramid_bytes(1, 64, 80, 9) == 0


def test_product_fails_loudly_without_gpu(rpe):
    import torch
    from rpe_amd import ops
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(rpe.RpeError):
        ops.se3_exp(torch.zeros(1, 6))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'robust-pose-estimator_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


def test_bad_arguments_return_status_without_a_gpu(rpe):
    """Error behaviour of the boundary: null pointers / bad sizes give RPE_E_BADARG (-1) before anything touches the
    device, so this runs on the CPU-only build container too."""
    import ctypes
    L = rpe.lib()
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)                      # never dereferenced: argument validation fails first
    assert L.rpe_se3_exp(null, one, 4, 1, null) == -1
    assert L.rpe_se3_exp(one, one, 4, 7, null) == -1                     # unknown dtype
    assert L.rpe_se3_exp(one, one, 0, 1, null) == 0                      # empty batch is fine
    assert L.rpe_pose_solve(*([null] * 9), 1, 8, 8, 0, 8, one, null, null, null, one, null) == -1
    assert L.rpe_pose_solve(*([one] * 9), 1, 8, 8, 5, 8, one, null, null, null, one, null) == -1      # unknown mode
    assert L.rpe_pose_solve_opts(*([one] * 9), 1, 8, 8, 0, 8, 1e-7, 1e-9, 0, one, null, null, null, one, null) == -1   # history 0
    assert L.rpe_corr_lookup(one, one, 1, 8, 8, 4, 3, one, null) == -1                                 # radius != 4
    assert L.rpe_corr_build(one, one, 1, 250, 8, 8, 4, one, null) == -1                                # channels % 16
    assert L.rpe_depth_backproject_warp(*([one] * 9), 1, 12, 16, *([one] * 8), null) == -1             # h % 8
    assert L.rpe_bias_act(one, null, 1, 4, 16, 1, one, 3, 0, null, 0, 0, null) == -1                   # slice overflow
    # fused convolutions: descriptor validation and the layout of the ctypes mirror
    from rpe_amd import _lib
    assert ctypes.sizeof(_lib.ConvDesc) == 200
    assert L.rpe_conv_packed_floats(256, 256, 1, 5) == (16 * 5 + 1) * 16 * 256 and L.rpe_conv_packed_floats(126, 324, 1, 1) == (21 + 1) * 16 * 128
    assert L.rpe_conv_packed_floats(0, 4, 3, 3) == 0
    assert L.rpe_conv_fused(None, null) == -1 and L.rpe_conv_pack(null, 8, 8, 3, 3, one, null) == -1
    d = _lib.ConvDesc(x=16, packed=16, out=16, b=1, cin=16, cout=16, h=8, w=10, kh=3, kw=3, mode=0)
    assert L.rpe_conv_fused(ctypes.byref(d), null) == -3                  # width % 4 != 0
    d.w, d.kw = 12, 7
    assert L.rpe_conv_fused(ctypes.byref(d), null) == -3                  # kernel width 7
    d.kw, d.mode = 3, 2
    assert L.rpe_conv_fused(ctypes.byref(d), null) == -1                  # GATE_ZR without out2 / hidden


@pytest.mark.gpu
def test_c_abi_from_a_plain_hip_program(rpe, tmp_path):
    """No Python, no torch: tests/abi/abi_smoke.cpp links librpe_hip.so and calls the ABI on its own stream."""
    import subprocess
    from rpe_amd import _lib
    exe = str(tmp_path / 'abi_smoke')
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-I', os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'tests', 'abi', 'abi_smoke.cpp'), '-L', libdir, '-lrpe_hip', '-Wl,-rpath,' + libdir, '-o', exe],
                   check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'ABI_SMOKE_OK' in r.stdout, r.stdout + r.stderr
