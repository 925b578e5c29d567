import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _usable_cores():
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:                                          # the GPU box shows 256 CPUs but its cgroup grants 16
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


torch.set_num_threads(_usable_cores())            # the CPU oracle is ~100x slower when oversubscribed


def pytest_addoption(parser):
    parser.addoption('--conv-bf16x3', action='store_true', default=False,
                     help='run the suite with raft.CONV_BF16X3 = True (the labelled bf16x3 variant: the >= 128-channel 3x3 layers of the update block, '
                          'the 1x1 layers and the correlation build; the GRU keeps its f32 kernels unless a test sets raft.X3_GRU): every parity '
                          'test must pass unchanged under the switch')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    if config.getoption('--conv-bf16x3'):
        import rpe_amd  # noqa: F401
        from rpe_amd import raft
        raft.CONV_BF16X3 = True


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name))
    return {k: (torch.from_numpy(g[k]) if g[k].ndim > 0 else g[k]) for k in g.files}


SOLVER_KEYS = ['flow', 'pcl1', 'pcl2', 'w1', 'w2', 'mask1', 'mask2', 'K', 'loss_weight']


@pytest.fixture(scope='session')
def rpe():
    import rpe_amd
    from rpe_amd import ops  # noqa: F401
    return rpe_amd
