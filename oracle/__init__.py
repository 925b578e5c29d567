"""CPU oracle for the per-frame stereo pose solve.

TEST INFRASTRUCTURE ONLY.  Everything under ``oracle/`` is a plain PyTorch-CPU / numpy
restatement of the reference algorithm (aimi-lab/robust-pose-estimator), written so that the
HIP path can be checked against it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
(``robust-pose-estimator_amd``) never does and fails loudly if its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * solver / objective / masks / normalisation / L-BFGS closure: PINNED.  ``oracle/gen_golden.py``
    imports the reference's own ``core/pose/pose_head.py`` and ``core/geometry/pinhole_transforms.py``
    (in the build container only) and runs them with the real ``torch.optim.LBFGS``; outputs are
    committed under ``tests/golden/`` and ``oracle.pose_head`` reproduces them.
  * flow warps (``remap_from_flow`` / ``remap_from_flow_nearest``), ``skewmat``, ATE/RPE metrics:
    PINNED against the reference files imported unchanged.
  * lietorch SE3 exp/log/act/mul/inv: lietorch is an unpinned pip-from-git dependency that is not
    installed here (reference README.md:37).  Restated from its published algorithm
    (quaternion exp/log with Taylor guards, left Jacobian); pinned through the reference's own
    tests (tests/unit_test_pinhole_transforms.py:24-53, tests/unit_test_pose_head.py:38-50) and against scipy's
    matrix exponential / logarithm / Rotation as an independent statement of the same conventions (1e-12).
  * TinyUNet, PoseNet.infer / flow2depth / get_weight_maps / proj, PoseEstimator (f2f): PINNED (round 3) by
    ``gen_golden.py::gen_modules`` running the reference's own core/unet/unet.py, core/pose/pose_net.py and
    core/pose/pose_estimator.py (with oracle.raft.RAFT standing in for the empty submodule).
  * RAFT (core/RAFT submodule is empty in the reference checkout): restated from the published
    princeton-vl/RAFT architecture; PARITY UNPINNED (no reference test touches RAFT).
"""
