#!/usr/bin/env python3
"""Throughput benchmark of the per-frame stereo pose solve on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the whole hot path (``PoseNet.infer`` of the reference, core/pose/pose_net.py:60-85)
over one batch of ``--batch`` synthetic 640x512 stereo frame pairs per GPU: batch-2B RAFT with 12 GRU
iterations (HIP correlation build + lookup, HIP gates, HIP convex up-sampling), fused HIP depth /
back-projection / warp, the two TinyUNet heads, and the 8-iteration device-resident SE(3) solve.  Inputs are
resident in HBM before the timed region; weights are seeded random-init (the reference's checkpoint is a
stripped blob).  Frames shard across ranks with no data-path collective (weak scaling); the only collectives
are the timing barrier and the MAX over ranks.

Rank 0 prints ONE JSON line: metric/value (pose solves per second over the whole job), ``roofline`` for the
correlation-lookup kernel (algorithmic bytes per launch / HIP-event time measured inside the timed region) and
``roofline_conv`` (the fused f32-MFMA convolutions that now hold most of the step) and
``cpu_baseline`` (the CPU oracle -- a PyTorch-CPU port of the reference path -- on a bounded sample of the
same inputs with the same weights).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (guide: MI355X_MICROARCH.md); ~6.3 TB/s achievable


F32_MFMA_PEAK_TFLOPS = 157.3    # MI355X dense f32 matrix peak (256 CUs x 256 FLOP/clk x 2.4 GHz); the packed-f32 vector pipe shares it


def conv_roofline(events, steps):
    """k_conv_igemm (update-block + encoder residual-block convolutions, epilogues included): algorithmic FLOPs
    2*cin*kh*kw*cout per output element over the HIP-event time of the same launches."""
    if not events:
        return None
    ms = sum(a.elapsed_time(b) for a, b, _ in events)
    flop = sum(f for _, _, f in events)
    tf = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {'kernel': 'k_conv_igemm', 'bound': 'mfma', 'achieved': tf, 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': tf / F32_MFMA_PEAK_TFLOPS, 'launches_per_step': len(events) // max(1, steps), 'ms_per_step': ms / max(1, steps),
            'tflop_per_step': flop / max(1, steps) / 1e12}


def lookup_algorithmic_bytes(pairs, h8, w8, levels=4, r=4):
    """N_q * L * [(2r+2)^2 + (2r+1)^2] * 4 B + coords N_q * 8 B, per pair per launch."""
    nq = h8 * w8
    return pairs * (nq * levels * ((2 * r + 2) ** 2 + (2 * r + 1) ** 2) * 4 + nq * 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=16, help='frame pairs per step per GPU (RAFT batch = 2x)')
    ap.add_argument('--height', type=int, default=512)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--raft-iters', type=int, default=12)
    ap.add_argument('--solver', default='lbfgs', choices=['lbfgs', 'gn'])
    ap.add_argument('--solver-iters', type=int, default=8)
    ap.add_argument('--cpu-frames', type=int, default=8, help='frames timed on the CPU oracle (0 = skip)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))   # RCCL on ROCm
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    import rpe_amd  # noqa: F401  (raises if librpe_hip.so is missing: no fallback)
    from rpe_amd import pose_head, pose_net, synth  # noqa: F401

    H, W, B = args.height, args.width, args.batch
    cfg = synth.model_config(H, W, iters=args.raft_iters, lbgfs_iters=args.solver_iters, solver=args.solver)
    model = pose_net.PoseNet(cfg)
    synth.init_synthetic_weights(model, seed=1234)
    model.eval().to(dev)
    frames = synth.stereo_frames(seed=1000 + rank, n=B, h=H, w=W)
    gpu_in = {k: v.to(dev) for k, v in synth.infer_args(frames).items()}
    mask2_init = gpu_in['mask2'].clone()

    # HIP-event timing of the correlation lookup inside the timed region (the kernel runs on torch's current stream)
    lookup_events = []
    raft = model.flow
    real_lookup = rpe_amd.ops.CorrPyramid.lookup
    timing = {'on': False}

    def timed_lookup(self, coords, out=None):
        if not timing['on']:
            return real_lookup(self, coords, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_lookup(self, coords, out)
        e1.record()
        lookup_events.append((e0, e1))
        return r

    rpe_amd.ops.CorrPyramid.lookup = timed_lookup

    solve_events = []
    real_solve = rpe_amd.ops.pose_solve

    def timed_solve(*a, **k):
        if not timing['on']:
            return real_solve(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_solve(*a, **k)
        e1.record()
        solve_events.append((e0, e1))
        return r

    rpe_amd.ops.pose_solve = timed_solve
    rpe_amd.pose_head.ops.pose_solve = timed_solve

    # the fused convolutions (k_conv_igemm): FLOPs and HIP-event time of every launch inside the timed region
    conv_events = []
    real_conv = rpe_amd.ops.conv_fused

    def timed_conv(x, pc, *a, **k):
        if k.get('prepare'):                       # the GRU loop's prepared launchers: time every launch of them
            launch = real_conv(x, pc, *a, **k)
            flop = 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * pc.cin * pc.cout * pc.kh * pc.kw

            def timed_launch():
                if not timing['on']:
                    return launch()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = launch()
                e1.record()
                conv_events.append((e0, e1, flop))
                return r
            return timed_launch
        if not timing['on']:
            return real_conv(x, pc, *a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_conv(x, pc, *a, **k)
        e1.record()
        st = k.get('stride', 1)
        conv_events.append((e0, e1, 2.0 * x.shape[0] * (x.shape[2] // st) * (x.shape[3] // st) * pc.cin * pc.cout * pc.kh * pc.kw))
        return r

    rpe_amd.ops.conv_fused = timed_conv            # raft.py calls it as ops.conv_fused

    def step():
        gpu_in['mask2'].copy_(mask2_init)          # infer() mutates mask2 in place, as the reference does
        return model.infer(**gpu_in, ret_details=True)

    def barrier():
        if distributed:
            dist.barrier()

    out = step()                                   # set-up pass (untimed, not a warm-up step): MIOpen picks its
    torch.cuda.synchronize()                       # conv algorithms on first use, like a compile step
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    timing['on'] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing['on'] = False
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        pose, _, depth2, weights, time_flow, stereo_flow2 = out
        info = model.pose_head.problem.last_info.cpu()
        lk_ms = [a.elapsed_time(b) for a, b in lookup_events]
        lk_avg_s = sum(lk_ms) / max(1, len(lk_ms)) / 1e3
        alg = lookup_algorithmic_bytes(2 * B, H // 8, W // 8)
        achieved = alg / lk_avg_s / 1e9 if lk_avg_s > 0 else 0.0
        traffic = None                              # HBM bytes per launch from separate rocprofv3 --pmc passes
        try:                                        # (profiles/pmc_traffic.json), valid for the default workload only
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))['k_corr_lookup']
            if (B, H, W) == (16, 512, 640):
                traffic = pmc['traffic_bytes_per_launch']
        except (OSError, KeyError, ValueError):
            pass
        res = {
            'metric': 'stereo-pair pose solves/sec (640x512, 8 solver iters)',
            'value': world * B * args.steps / elapsed,
            'unit': 'pose solves/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / max(1, args.steps),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (RAFT / geometry) + f64 (SE(3) solve), as the reference',
            'data': 'synthetic (seeded rendered stereo pairs, seeded random-init weights)',
            'config': {'workload': f'PoseNet.infer, {W}x{H} stereo frame pairs, {B} per GPU per step (RAFT batch {2 * B}), '
                                   f'{args.raft_iters} GRU iters, {args.solver} x{args.solver_iters} SE(3) solve, weight heads on',
                       'frames_per_gpu': B, 'height': H, 'width': W, 'raft_iters': args.raft_iters,
                       'solver': args.solver, 'solver_iters': args.solver_iters, 'parallelism': f'frames sharded x{world}'},
            'roofline': {'kernel': 'k_corr_lookup', 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'algorithmic_bytes_per_launch': alg,
                         'avg_launch_us': lk_avg_s * 1e6, 'launches_timed': len(lk_ms)},
            'roofline_pose_solve': pose_roofline(solve_events, B, H, W, args.solver_iters),
            'roofline_conv': conv_roofline(conv_events, args.steps),
            'solver_iters_run': {'min': int(info[:, 0].min()), 'max': int(info[:, 0].max())},
            'valid_fraction': float(gpu_in['mask2'].float().mean()),
            'peak_hbm_gb': torch.cuda.max_memory_allocated(dev) / 1e9,
        }
        if args.cpu_frames > 0 and world == 1:        # CPU baseline on rank 0 at N = 1 only
            res['cpu_baseline'] = cpu_baseline(cfg, model, frames, args.cpu_frames, pose)
        print(json.dumps(res))
    barrier()
    if distributed:
        dist.destroy_process_group()


def pose_roofline(events, frames, h, w, iters):
    """Second HBM-bound kernel family: the whole device-resident solve (N x (k_pose_reduce + k_pose_update)), HIP-event
    timed in the timed region; algorithmic bytes = 42 B/pixel per evaluation."""
    if not events:
        return None
    t = sum(a.elapsed_time(b) for a, b in events) / len(events) / 1e3
    alg = frames * h * w * 42 * iters
    return {'kernels': 'k_pose_reduce + k_pose_update x%d' % iters, 'bound': 'hbm', 'achieved': alg / t / 1e9, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': alg / t / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_solve': alg, 'avg_solve_us': t * 1e6}


def usable_cores():
    """Host threads this process may really use: scheduler affinity capped by the cgroup CPU quota (the GPU box
    shows 256 logical CPUs but grants 16; 256 threads on 16 CPUs is ~100x slower than 16 threads)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(cfg, model, frames, n_frames, gpu_pose):
    """The CPU oracle (PyTorch-CPU port of the reference path, oracle/pose_net.py) on the first frames of the
    same batch with the same weights, all host threads.  Frames are processed one at a time, as the reference
    does (scripts/infer_trajectory.py:57 batch_size=1)."""
    from oracle import pose_net as opn
    from rpe_amd import synth
    cores = usable_cores()
    torch.set_num_threads(cores)
    om = opn.PoseNet(cfg)
    om.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    om.eval()
    n_frames = min(n_frames, frames['image1l'].shape[0])
    poses = []
    t0 = time.perf_counter()
    for i in range(n_frames):
        a = {k: v[i:i + 1].clone() for k, v in synth.infer_args(frames).items()}
        poses.append(om.infer(**a))
    dt = time.perf_counter() - t0
    gp = gpu_pose.data.reshape(-1, 7)[:n_frames].cpu()
    diff = float((torch.cat(poses).reshape(-1, 7) - gp).abs().max())
    return {'value': n_frames / dt, 'unit': 'pose solves/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n_frames} frame pairs of the same batch, one at a time, torch CPU f32 + f64 L-BFGS',
            'max_abs_pose_diff_vs_gpu': diff}


if __name__ == '__main__':
    main()
