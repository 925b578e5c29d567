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
