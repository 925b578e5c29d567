"""MI355X-native hot path of aimi-lab/robust-pose-estimator's per-frame stereo pose solve.

The directory name carries a hyphen (it mirrors the reference repository's name); import it through the
root-level alias module:  ``import rpe_amd``.

Layout:
  csrc/               hand-written HIP kernels for gfx950 + the C ABI (include/rpe.h) -> librpe_hip.so
  _lib.py, ops.py     ctypes binding and tensor-level wrappers (no CPU fallback)
  se3.py              lietorch-style SE3 value type on the rpe_se3_* kernels
  raft.py, unet.py    RAFT / TinyUNet host mirrors (convolutions on PyTorch-ROCm, correlation / gates /
                      up-sampling in HIP)
  pose_head.py        DPoseSE3Head / DeclarativeLayerLie mirrors on rpe_pose_solve
  pose_net.py         PoseNet.infer / flow2depth mirror
  pose_estimator.py   frame-to-frame tracker mirror (+ Frame)
  sharding.py         one-process-per-GPU sequence sharding, RCCL all-gather of relative poses
  synth.py            seeded synthetic stereo inputs for tests and bench
"""
__version__ = '0.1.0'

from . import _lib  # noqa: F401
from ._lib import RpeError, build, lib  # noqa: F401
