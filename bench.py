#!/usr/bin/env python3
"""Throughput benchmark of the per-frame stereo pose solve on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts N rank processes itself, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               (the driver's form: one rank per GPU over RCCL)
    python bench.py --mode sequence [--gpus N]               (sharded tracker over a synthetic sequence, incl. the all-gather)

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

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (guide: MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0    # what streaming kernels reach on this part (same guide; tools/hbm_roof_probe.py measured 5.3-5.8 TB/s for copy / add)


CONV_PASS_NOTE = 'a diagnostic pass of 2 steps after the timed region (HIP events around every convolution launch; the timed region itself brackets only the lookup and the solve)'
F32_MFMA_PEAK_TFLOPS = 157.3    # MI355X dense f32 matrix peak (256 CUs x 256 FLOP/clk x 2.4 GHz); the packed-f32 vector pipe shares it


def conv_roofline(events, steps):
    """k_conv_igemm (update-block + encoder residual-block convolutions, epilogues included): algorithmic FLOPs
    2*cin*kh*kw*cout per output element over the HIP-event time of the same launches."""
    if not events:
        return None
    ms = sum(a.elapsed_time(b) for a, b, _ in events)
    flop = sum(f for _, _, f in events)
    tf = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {'kernel': 'k_conv_igemm + k_conv1x1', 'bound': 'mfma', 'achieved': tf, 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': tf / F32_MFMA_PEAK_TFLOPS, 'launches_per_step': len(events) // max(1, steps), 'ms_per_step': ms / max(1, steps),
            'tflop_per_step': flop / max(1, steps) / 1e12, 'measured_in': CONV_PASS_NOTE}


def wino_roofline(events, steps, kernel='k_conv_wino'):
    """k_conv_wino1d (the GRU's 1x5 / 5x1 convolutions as Winograd F(4,5) along the axis, gate epilogues included) and k_conv_wino (the update block's four 3x3 layers and the encoders' stride-1 layers as Winograd F(2x2,3x3)): EXECUTED matrix FLOPs over HIP-event time
    against the f32 MFMA peak (so frac <= 1); ``effective`` = the direct convolution's FLOPs over the same time."""
    if not events:
        return None
    ms = sum(a.elapsed_time(b) for a, b, _, _ in events)
    ex = sum(f for _, _, f, _ in events)
    eff = sum(f for _, _, _, f in events)
    tf = ex / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {'kernel': kernel, 'bound': 'mfma', 'achieved': tf, 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': tf / F32_MFMA_PEAK_TFLOPS, 'effective_tflops_direct_equivalent': eff / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
            'launches_per_step': len(events) // max(1, steps), 'ms_per_step': ms / max(1, steps),
            'executed_tflop_per_step': ex / max(1, steps) / 1e12, 'measured_in': CONV_PASS_NOTE}


def lookup_algorithmic_bytes(pairs, h8, w8, levels=4, r=4):
    """N_q * L * [(2r+2)^2 + (2r+1)^2] * 4 B + coords N_q * 8 B, per pair per launch."""
    nq = h8 * w8
    return pairs * (nq * levels * ((2 * r + 2) ** 2 + (2 * r + 1) ** 2) * 4 + nq * 8)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--mode', default='batch', choices=['batch', 'sequence'],
                    help='batch: PoseNet.infer on --batch frame pairs per GPU per step (the headline metric); '
                         'sequence: the sharded frame-to-frame tracker over a synthetic sequence, all-gather included')
    ap.add_argument('--batch', type=int, default=16, help='frame pairs per step per GPU (RAFT batch = 2x)')
    ap.add_argument('--height', type=int, default=512)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--raft-iters', type=int, default=12)
    ap.add_argument('--solver', default='lbfgs', choices=['lbfgs', 'gn'])
    ap.add_argument('--solver-iters', type=int, default=8)
    ap.add_argument('--cpu-frames', type=int, default=8, help='frames timed on the CPU oracle (0 = skip)')
    ap.add_argument('--seq-frames', type=int, default=33, help='--mode sequence: frames per GPU (weak scaling); 33 = 32 pairs = two chunks of 16 at N = 1')
    ap.add_argument('--seq-chunk', type=int, default=16, help='--mode sequence: frames per RAFT pass (SequenceTracker(chunk=...)); 1 = one frame at a time')
    ap.add_argument('--fp16-features', action='store_true',
                    help="BASELINE config 5: fp16 feature maps into the correlation (upstream RAFT's mixed_precision), f32 pyramid, f64 solve")
    ap.add_argument('--corr-bf16x3', action='store_true',
                    help='EXPERIMENT (reported under its own dtype, never the headline): the correlation build with every f32 product as six bf16 products of an exact 3-way split (RPE_F32X3)')
    ap.add_argument('--conv-bf16x3', action='store_true',
                    help='LABELLED VARIANT (reported under its own dtype, never the headline): the update block\'s 3x3 layers with >= 128 input channels, the '
                         '1x1 layers (convc1, the encoders\' and the mask head\'s output layers) and the correlation build with every f32 product as six bf16 '
                         'products of an exact 3-way split on the 16-bit matrix cores (raft.CONV_BF16X3)')
    ap.add_argument('--no-extras', action='store_true', help='skip the batch-1 latency / tracker / Gauss-Newton lines (and the live PMC traffic passes)')
    ap.add_argument('--one-stream', action='store_true', help='encoders one after the other on one stream (raft.ENC_STREAMS = False): for kernel-trace profiles whose per-kernel durations must not overlap')
    ap.add_argument('--no-live-traffic', action='store_true',
                    help='do not run the two rocprofv3 --pmc child passes after the timed region; roofline.traffic then comes from profiles/pmc_traffic.json')
    return ap.parse_args(argv)


X3_DTYPE = ('f32 (RAFT / geometry; LABELLED VARIANT: f32 products of the update block\'s 3x3 layers with >= 128 input channels, of the 1x1 layers and of the correlation build '
            'as six bf16 products of an exact 3-way split, f32 accumulation) + f64 (SE(3) solve)')
X3_FAMILIES = {'split (bf16x3)': ['k_conv_wino_x3: BasicMotionEncoder.convc2, .conv, FlowHead.conv1, mask head 3x3', 'k_conv1x1_x3: BasicMotionEncoder.convc1, the 1x1 output layers (fnet, cnet ReLU half, mask head)',
                                  'k_corr_build_x3: correlation pyramid'],
               'f32 matrix cores (not split: measured no faster)': ['k_conv_wino: encoders\' 3x3 layers, convf2',
                                                                    'k_conv_wino1d: SepConvGRU 1x5 / 5x1 (k_conv_wino1d_x3 exists, raft.X3_GRU: 449 / 410 / 251 / 237 us per launch in this step '
                                                                    'against 407 / 395 / 240 / 224)',
                                                                    'k_conv1x1 / k_conv_igemm: cnet tanh half, stride-2 layers', 'k_stem7x7: stems']}


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start N rank processes (one per GPU, RCCL over xGMI) through
    torch.distributed.run as a CHILD process and relay its exit code.  The parent never touches the GPU
    (``device_count`` does not initialise HIP on this image), so nothing is exec'ed over an initialised runtime."""
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus:
        print(f'bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible', file=sys.stderr)
        return 2
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU', file=sys.stderr)
        sys.exit(2)
    import torch.distributed as dist
    distributed = world > 1 or args.mode == 'sequence'      # sequence mode runs its all-gather under RCCL even at N = 1
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    rccl_ranks = None
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            os.environ['MASTER_PORT'] = str(free_port())
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)   # backend "nccl" IS RCCL on ROCm
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                              # a real collective: every rank must have joined
        rccl_ranks = int(ones.item())

    import rpe_amd  # noqa: F401  (raises if librpe_hip.so is missing: no fallback)
    if args.one_stream:
        from rpe_amd import raft as _raft
        _raft.ENC_STREAMS = False
    if args.corr_bf16x3:
        from rpe_amd import raft as _raft
        _raft.CORR_BF16X3 = True                        # the labelled experiment (its own dtype string below)
    if args.conv_bf16x3:
        from rpe_amd import raft as _raft
        _raft.CONV_BF16X3 = True                        # the labelled variant (its own dtype string below)
    if args.mode == 'sequence':
        res = run_sequence(args, rank, world, dev, dist)
    else:
        res = run_batch(args, rank, world, dev, dist if distributed else None)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        res['rccl_ranks'] = rccl_ranks
        # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio, which would otherwise be
        # flushed at exit, after Python's line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(res), flush=True)


RANK_MS = {}          # per-rank ms/step (min / max over ranks) of the last timed_region


def timed_region(step, steps, warmup, dev, dist):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
    def barrier():
        if dist is not None:
            dist.barrier()
    def sync():
        if torch.device(dev).type == 'cuda':             # (the gloo tests drive the sequence pass on CPU tensors)
            torch.cuda.synchronize()
    out = None
    for _ in range(warmup):
        out = step()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync()
    busy = time.perf_counter() - t0                  # this rank's own work, without waiting for the others
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    own = busy / max(1, steps)
    RANK_MS.clear()
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # each rank's own time per step BEFORE the closing barrier (the MAX-reduced value hides which rank is slow)
        mine = torch.tensor([own], device=dev, dtype=torch.float64)
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        RANK_MS.update(min=1e3 * float(lo), max=1e3 * float(hi))
    else:
        RANK_MS.update(min=1e3 * own, max=1e3 * own)
    return elapsed, out


def sequence_pass(track, n_frames, rank, steps, warmup, dev, dist):
    """K timed walks of a sharded sequence: ``track()`` -> (poses (F,7), rel, ok) on every rank (sharding.SequenceTracker.track or
    sharding.track_sharded: each rank's block, ONE all-gather of the relative poses, gate + prefix product), then the check that every
    rank holds the SAME trajectory (MIN and MAX over ranks of a checksum agree).  Used by --mode sequence and, when world > 1, by the
    batch mode's extra line (the number that exercises the collective); the gloo test drives it on the CPU oracle."""
    elapsed, (poses, rel, ok) = timed_region(track, steps, warmup, dev, dist)
    chk = torch.nan_to_num(poses.double()).mul(torch.arange(1, poses.numel() + 1, device=poses.device, dtype=torch.float64).reshape(poses.shape)).sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = bool(float(lo) == float(hi))
    if not same:
        print(f'bench.py sequence pass: rank {rank}: trajectory checksums differ across ranks ({float(lo)!r} .. {float(hi)!r})', file=sys.stderr)
    return {'elapsed': elapsed, 'frames_per_s': n_frames * steps / elapsed if elapsed > 0 else 0.0, 'poses': poses, 'rel': rel, 'ok': ok, 'same': same,
            'checksum': float(chk)}


def sequence_tracker(args, rank, world, dev, frames_per_gpu):
    """(SequenceTracker, total frames) over synthetic frames sharded in contiguous blocks with a one-frame halo, this rank's frames in HBM."""
    from rpe_amd import pose_estimator, pose_net, sharding, synth
    H, W = args.height, args.width
    F = frames_per_gpu * world
    cfg = synth.model_config(H, W, iters=args.raft_iters, lbgfs_iters=20, solver=args.solver)   # infer_f2f.yaml:11
    model = synth.init_synthetic_weights(pose_net.PoseNet(cfg), seed=1234).eval().to(dev)
    slam = dict(frame2frame=True, depth_clipping=[1, 250], lbgfs_iters=20, conf_weighing=True)
    s, e = sharding.block_partition(F - 1, world)[rank]
    cache = {}
    for t in range(s, e + 1):                               # this rank's frames incl. the halo, resident in HBM
        fr = synth.stereo_frames(seed=5000 + t, n=1, h=H, w=W)
        cache[t] = (fr['image2l'].to(dev), fr['image2r'].to(dev), fr['mask2'].to(dev))
    K = synth.intrinsics(H, W)
    make = lambda: pose_estimator.PoseEstimator(slam, K, 7.2 * 250.0, model, (W, H)).to(dev)
    get = lambda t: (cache[t][0], cache[t][1], cache[t][2].clone())
    return sharding.SequenceTracker(make, get, chunk=args.seq_chunk), F


def run_sequence(args, rank, world, dev, dist):
    """BASELINE config 4: a stereo sequence sharded over the ranks in contiguous blocks with a one-frame halo
    (rpe_amd.sharding, scripts/infer_trajectory.py:57,71-97 + core/pose/pose_estimator.py:81-91 of the reference),
    every rank walking its block in chunks of --seq-chunk frames, ONE all-gather of the relative poses, then gate + prefix product.
    A step = tracking the whole sequence once; value = frames/s over the whole job."""
    H, W, Fg = args.height, args.width, args.seq_frames
    tracker, F = sequence_tracker(args, rank, world, dev, Fg)
    import warnings

    def step():
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')                 # unrelated synthetic frames: many pairs fail the |log| gate
            return tracker.track(F, rank, world)
    seq = sequence_pass(step, F, rank, args.steps, args.warmup, dev, dist)
    elapsed, poses, ok, same, chk = seq['elapsed'], seq['poses'], seq['ok'], seq['same'], seq['checksum']
    if not same:                                            # a sharded run whose ranks disagree about the trajectory is a failed run
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(3)
    if rank != 0:
        return None
    return {
        'metric': 'sequence tracking frames/sec (640x512, frame-to-frame, L-BFGS 20), frames sharded over ranks + RCCL all-gather',
        'value': F * args.steps / elapsed, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / max(1, args.steps), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 (RAFT / geometry) + f64 (SE(3) solve), as the reference',
        'data': 'synthetic (independent seeded stereo frames used as a sequence, seeded random-init weights)',
        'config': {'workload': f'SequenceTracker.track, {F} frames of {W}x{H} ({Fg} per GPU, contiguous blocks + 1-frame halo), '
                               f'chunks of {args.seq_chunk} frames (one RAFT pass per chunk; bit-identical to one frame at a time), '
                               f'{args.raft_iters} GRU iters, {args.solver} x20 solve, one all-gather of (frames,8) f32',
                   'frames': F, 'frames_per_gpu': Fg, 'chunk': args.seq_chunk, 'parallelism': f'sequence blocks x{world}'},
        'rank_ms_per_step': dict(RANK_MS),
        'poses_finite': bool(torch.isfinite(poses).all()), 'pairs_accepted': int(ok.sum()), 'poses_shape': list(poses.shape),
        'poses_equal_on_all_ranks': same, 'poses_checksum': chk,
    }


def run_batch(args, rank, world, dev, dist):
    import rpe_amd
    from rpe_amd import pose_head, pose_net, synth  # noqa: F401
    from rpe_amd import raft as raft_mod

    H, W, B = args.height, args.width, args.batch
    cfg = synth.model_config(H, W, iters=args.raft_iters, lbgfs_iters=args.solver_iters, solver=args.solver, mixed_precision=args.fp16_features)
    model = pose_net.PoseNet(cfg)
    synth.init_synthetic_weights(model, seed=1234)
    model.eval().to(dev)
    frames = synth.stereo_frames(seed=1000 + rank, n=B, h=H, w=W)
    gpu_in = {k: v.to(dev) for k, v in synth.infer_args(frames).items()}
    mask2_init = gpu_in['mask2'].clone()

    # HIP-event timing of the correlation lookup inside the timed region (the kernel runs on torch's current stream)
    lookup_events = []
    real_lookup = rpe_amd.ops.CorrPyramid.lookup
    timing = {'on': False, 'conv': False}        # 'on': lookup + solve events (the timed region); 'conv': every convolution launch (a separate diagnostic pass)

    last_lookup = {}

    def timed_lookup(self, coords, out=None, prepare=False):
        last_lookup['pyr'], last_lookup['coords'], last_lookup['out'] = self, coords, out      # (the loop's persistent buffers: coords1 / corr of the workspace)
        if prepare:
            return real_lookup(self, coords, out, prepare=True)
        if not timing['on']:
            return real_lookup(self, coords, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_lookup(self, coords, out)
        e1.record()
        lookup_events.append((e0, e1))
        return r

    rpe_amd.ops.CorrPyramid.lookup = timed_lookup
    # The product enqueues the update loop as ONE launch list (raft.LOOP_OPLIST -> rpe_run_ops), so the lookup is not a Python call any
    # more: the list itself records raw HIP events around each iteration's lookup when raft.LOOKUP_EVENT_SINK hands it some.  One pair per
    # lookup launch of the timed region, created up front (no event is created, and no extra record issued, inside the timed region).
    from rpe_amd import _lib as _rpe_lib
    raw_ev = _rpe_lib.RawEvents(2 * args.raft_iters * max(1, args.steps))
    raw_used = [0]

    def lookup_sink(iters):
        k = raw_used[0]
        if k + 2 * iters > len(raw_ev.handles):          # (more passes than planned: those run untimed)
            return [None] * (2 * iters)
        raw_used[0] = k + 2 * iters
        return raw_ev.handles[k:k + 2 * iters]

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

    # the fused convolutions (k_conv_igemm): FLOPs and HIP-event time of every launch -- in a diagnostic pass AFTER the timed region: ~300 event
    # records per step inside it cost the headline 1.5-2 % (68.3 vs 69.6 ms per step), and only the lookup's timing has to come from there
    conv_events, wino1d_events = [], []
    real_conv = rpe_amd.ops.conv_fused

    def timed_conv(x, pc, *a, **k):
        if k.get('prepare'):                       # the GRU loop's prepared launchers: time every launch of them
            launch = real_conv(x, pc, *a, **k)
            flop = 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * pc.cin * pc.cout * pc.kh * pc.kw
            wino1d = k.get('entry') == 'rpe_conv_wino1d'       # (ops.conv_wino1d goes through conv_fused): 8 products per 4 outputs, not 20

            def timed_launch():
                if not timing['conv']:
                    return launch()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = launch()
                e1.record()
                if wino1d:
                    wino1d_events.append((e0, e1, flop * 0.4, flop))
                else:
                    conv_events.append((e0, e1, flop))
                return r
            timed_launch.op, timed_launch.keep = launch.op, launch.keep      # (a launch list takes the real argument block: no Python in between)
            return timed_launch
        if not timing['conv']:
            return real_conv(x, pc, *a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_conv(x, pc, *a, **k)
        e1.record()
        st = k.get('stride', 1)
        flop = 2.0 * x.shape[0] * (x.shape[2] // st) * (x.shape[3] // st) * pc.cin * pc.cout * pc.kh * pc.kw
        if k.get('entry') == 'rpe_conv_wino1d':    # the GRU's context terms
            wino1d_events.append((e0, e1, flop * 0.4, flop))
        else:
            conv_events.append((e0, e1, flop))
        return r

    rpe_amd.ops.conv_fused = timed_conv            # raft.py calls it as ops.conv_fused

    # the Winograd 3x3 layers (k_conv_wino): EXECUTED matrix FLOPs = 16 products per 2x2 output tile and (ci, co) pair,
    # i.e. 4/9 of the direct form's; the direct-form count is kept separately as "effective"
    wino_events = []
    real_wino = rpe_amd.ops.conv_wino

    def timed_wino(x, pw, *a, **k):
        direct = 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * pw.cin * pw.cout * 9
        if not k.get('prepare'):                   # the encoders' layers: launched directly
            if not timing['conv']:
                return real_wino(x, pw, *a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = real_wino(x, pw, *a, **k)
            e1.record()
            wino_events.append((e0, e1, direct * 4.0 / 9.0, direct))
            return r
        launch = real_wino(x, pw, *a, **k)

        def timed_launch():
            if not timing['conv']:
                return launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = launch()
            e1.record()
            wino_events.append((e0, e1, direct * 4.0 / 9.0, direct))
            return r
        timed_launch.op, timed_launch.keep = launch.op, launch.keep
        return timed_launch

    rpe_amd.ops.conv_wino = timed_wino

    def step():
        gpu_in['mask2'].copy_(mask2_init)          # infer() mutates mask2 in place, as the reference does
        return model.infer(**gpu_in, ret_details=True)

    def timed_step():
        timing['on'] = True
        raft_mod.LOOKUP_EVENT_SINK = lookup_sink
        try:
            return step()
        finally:
            timing['on'] = False
            raft_mod.LOOKUP_EVENT_SINK = None

    step()                                         # set-up pass (untimed, not a warm-up step): weights are packed and the
    torch.cuda.synchronize()                       # persistent workspaces / launch descriptors built on first use
    for _ in range(args.warmup):
        step()
    elapsed, out = timed_region(timed_step, args.steps, 0, dev, dist)

    seq_line = None
    if world > 1 and dist is not None and not args.no_extras:
        # The batch metric shards frames with no data-path collective; the multi-GPU number that exercises the all-gather of relative poses
        # and the prefix product is a SEQUENCE walk.  A short one (one chunk of 16 + halo per GPU, one timed walk) on every rank, after
        # the timed region, so that a scaling run's line carries it.  (N > 1 has never been run on hardware in this pool.)
        import warnings
        tracker, Fs = sequence_tracker(args, rank, world, dev, 17)

        def seq_step():
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                return tracker.track(Fs, rank, world)
        sp = sequence_pass(seq_step, Fs, rank, 1, 1, dev, dist)
        seq_line = {'sequence_frames_per_s': sp['frames_per_s'], 'sequence_frames': Fs, 'sequence_poses_equal_on_all_ranks': sp['same'],
                    'sequence_config': f'SequenceTracker.track, {Fs} frames ({17} per GPU, contiguous blocks + 1-frame halo, chunks of {args.seq_chunk}), one all-gather of (frames,8) f32; 1 warm-up + 1 timed walk'}
        del tracker
    if rank != 0:
        return None
    pose, _, depth2, weights, time_flow, stereo_flow2 = out
    conv_steps = 2                                 # diagnostic pass: the same step with every convolution launch bracketed by HIP events
    timing['conv'] = True
    enc_streams, raft_mod.ENC_STREAMS = raft_mod.ENC_STREAMS, False     # one stream: a launch's duration is its own, not two overlapping launches'
    loop_oplist, raft_mod.LOOP_OPLIST = raft_mod.LOOP_OPLIST, False     # launch by launch from Python: the wrappers above bracket each launch with events
    try:
        for _ in range(conv_steps):
            step()
        torch.cuda.synchronize()
    finally:
        timing['conv'] = False
        raft_mod.ENC_STREAMS = enc_streams
        raft_mod.LOOP_OPLIST = loop_oplist
    info = model.pose_head.problem.last_info.cpu()
    lk_ms = sorted([a.elapsed_time(b) for a, b in lookup_events] + [raw_ev.elapsed_ms(i, i + 1) for i in range(0, raw_used[0], 2)])
    lk_avg_s = sum(lk_ms) / max(1, len(lk_ms)) / 1e3

    def pct(q):
        return lk_ms[min(len(lk_ms) - 1, int(q * len(lk_ms)))] * 1e3 if lk_ms else 0.0
    lk_med_s = pct(0.5) / 1e6
    alg = lookup_algorithmic_bytes(2 * B, H // 8, W // 8)
    # frac from the MEDIAN launch: a HIP-event pair also spans the launch gap in front of the kernel, and a mean over 240 launches carries
    # the outliers of the box (round 4: 98.3 us mean in the driver's run against 93.9 us by rocprofv3); mean, p10 and p90 are reported beside it
    achieved = alg / lk_med_s / 1e9 if lk_med_s > 0 else 0.0
    achieved_mean = alg / lk_avg_s / 1e9 if lk_avg_s > 0 else 0.0
    traffic = traffic_src = None                # HBM bytes per launch from separate rocprofv3 --pmc passes over THIS program
    try:                                        # (tools/pmc_bench.sh -> profiles/pmc_traffic.json), valid for the default workload only
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))['k_corr_lookup_bench']
        if (B, H, W) == (16, 512, 640):
            traffic, traffic_src = pmc['traffic_bytes_per_launch'], pmc['source']
    except (OSError, KeyError, ValueError):
        pass
    rounds_own = lookup_rounds(last_lookup['pyr'], last_lookup['coords'])      # roughness of the coordinates this run looked up last
    res = {
        'metric': 'stereo-pair pose solves/sec (640x512, 8 solver iters)',
        'value': world * B * args.steps / elapsed,
        'unit': 'pose solves/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / max(1, args.steps),
        'rank_ms_per_step': dict(RANK_MS),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': ('f32 (RAFT / geometry; fp16 feature maps into the correlation) + f64 (SE(3) solve)' if args.fp16_features
                  else 'f32 (RAFT / geometry; EXPERIMENT: correlation products as six bf16 products of an exact 3-way split) + f64 (SE(3) solve)' if args.corr_bf16x3
                  else X3_DTYPE if args.conv_bf16x3
                  else 'f32 (RAFT / geometry) + f64 (SE(3) solve), as the reference'),
        'data': 'synthetic (seeded rendered stereo pairs, seeded random-init weights)',
        'config': {'workload': f'PoseNet.infer, {W}x{H} stereo frame pairs, {B} per GPU per step (RAFT batch {2 * B}), '
                               f'{args.raft_iters} GRU iters, {args.solver} x{args.solver_iters} SE(3) solve, weight heads on'
                               + (', fp16 features' if args.fp16_features else '') + (', correlation bf16x3 (experiment)' if args.corr_bf16x3 else '')
                               + (', conv + correlation bf16x3 (labelled variant)' if args.conv_bf16x3 else ''),
                   'frames_per_gpu': B, 'height': H, 'width': W, 'raft_iters': args.raft_iters,
                   'solver': args.solver, 'solver_iters': args.solver_iters, 'parallelism': f'frames sharded x{world}'},
        'roofline': {'kernel': 'k_corr_lookup', 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'peak_achievable': HBM_ACHIEVABLE_GBS, 'frac_of_achievable': achieved / HBM_ACHIEVABLE_GBS,
                     'traffic': traffic, 'algorithmic_bytes_per_launch': alg,
                     'median_launch_us': lk_med_s * 1e6, 'p10_launch_us': pct(0.1), 'p90_launch_us': pct(0.9), 'avg_launch_us': lk_avg_s * 1e6,
                     'frac_from_mean': achieved_mean / HBM_PEAK_GBS, 'frac_from_median': achieved / HBM_PEAK_GBS,
                     'frac_definition': 'frac = algorithmic bytes / MEDIAN launch duration since round 5 (rounds 1-4 used the mean: compare frac_from_mean)',
                     'launches_timed': len(lk_ms), 'traffic_source': traffic_src,
                     'timed_by': 'raw HIP events recorded by the launch list itself around each lookup of the timed region (rpe_run_ops, RPE_OP_EVENT_RECORD)',
                     'coordinates': 'the final GRU iteration of this run (random-init RAFT: near-uniform drift)', **rounds_own},
        'roofline_pose_solve': pose_roofline(solve_events, B, H, W, args.solver_iters),
        'roofline_conv': conv_roofline(conv_events, conv_steps),
        'roofline_conv_winograd_1d': wino_roofline(wino1d_events, conv_steps, 'k_conv_wino1d'),
        'roofline_conv_winograd': wino_roofline(wino_events, conv_steps),
        'solver_iters_run': {'min': int(info[:, 0].min()), 'max': int(info[:, 0].max())},
        'valid_fraction': float(gpu_in['mask2'].float().mean()),
        'peak_hbm_gb': torch.cuda.max_memory_allocated(dev) / 1e9,
    }
    # the whole step against the f32 matrix peak: executed FLOPs of the matrix kernels (Winograd layers at their executed 4/9 and 2/5)
    # over the step time -- the floor the f32 formulation could reach at 100 % of the pipe (6.09 TFLOP, 38.7 ms, 0.58)
    fam = [res['roofline_conv'], res['roofline_conv_winograd'], res['roofline_conv_winograd_1d']]
    ex = sum((f.get('tflop_per_step') or f.get('executed_tflop_per_step') or 0.0) for f in fam if f)
    ex += 2.0 * (2 * B) * float((H // 8) * (W // 8)) ** 2 * 256 / 1e12                      # correlation build: 2 N_q^2 C per pair
    res['roofline_step'] = {'executed_tflop': ex, 'floor_ms_at_peak': ex / F32_MFMA_PEAK_TFLOPS * 1e3,
                            'frac': (ex / F32_MFMA_PEAK_TFLOPS * 1e3) / res['ms_per_step'] if res['ms_per_step'] > 0 else 0.0,
                            'counts': 'k_conv_igemm + k_conv1x1 + k_conv_wino + k_conv_wino1d (executed) + k_corr_build; stems, heads and element-wise passes not counted'}
    if args.conv_bf16x3:
        res['conv_bf16x3_families'] = X3_FAMILIES
    if seq_line is not None:
        res.update(seq_line)
    if world == 1 and not args.no_extras and not args.conv_bf16x3 and not args.corr_bf16x3 and not args.fp16_features:
        # the labelled bf16x3 variant of the same step, measured AFTER the timed region (the headline above is pure f32)
        raft_mod.CONV_BF16X3 = True
        try:
            step(); torch.cuda.synchronize()                 # set-up pass: the variant's packings and launch descriptors
            x3_steps = max(3, min(10, args.steps))
            t0 = time.perf_counter()
            for _ in range(x3_steps):
                out_x3 = step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            dpose = (out_x3[0].vec() - pose.vec()).abs().max().item() if hasattr(pose, 'vec') else None
            res['value_conv_bf16x3'] = {'value': B * x3_steps / dt, 'unit': 'pose solves/s', 'ms_per_step': 1e3 * dt / x3_steps, 'steps': x3_steps,
                                        'dtype': X3_DTYPE, 'families': X3_FAMILIES, 'max_abs_pose_diff_vs_f32_step': dpose,
                                        'note': 'labelled variant, not the headline: bench.py --conv-bf16x3 runs the whole contract on it'}
        except Exception as e:                               # an extra must never cost the already-measured headline line
            res['value_conv_bf16x3'] = {'error': f'{type(e).__name__}: {e}'}
        finally:
            raft_mod.CONV_BF16X3 = False
        try:
            step(); torch.cuda.synchronize()                 # back on the f32 packings for the passes below
        except Exception as e:
            res['value_conv_bf16x3_restore_error'] = f'{type(e).__name__}: {e}'
    if world == 1 and not args.no_extras and not args.no_live_traffic:
        live = live_lookup_traffic(args)              # HBM bytes per lookup launch measured NOW, on this box, over this program's own launches
        if live is not None:
            res['roofline'].update(traffic=live['traffic_bytes_per_launch'], traffic_source=live['source'], traffic_detail=live)
    if world == 1 and not args.no_extras:             # deployment numbers, measured after the timed region on rank 0
        res['roofline_lookup_realistic'] = lookup_realistic(last_lookup['pyr'], last_lookup['out'], B, H, W, dev)
        res.update(deployment_numbers(args, model, cfg, frames, gpu_in, mask2_init, dev, solve_events, timing))
    if args.cpu_frames > 0 and world == 1:            # CPU baseline on rank 0 at N = 1 only
        res['cpu_baseline'] = cpu_baseline(cfg, model, frames, args.cpu_frames, pose)
    return res


def live_lookup_traffic(args):
    """roofline.traffic measured in THIS run: two child passes of this very program under ``rocprofv3 --pmc <one counter> --kernel-trace``
    (FETCH_SIZE, then WRITE_SIZE: separate passes, --kernel-trace only, the program itself after ``--``, as MI355X_MICROARCH.md's HBM
    section prescribes), after the timed region so that they disturb nothing.  Bytes per k_corr_lookup launch = 2 x FETCH_SIZE KiB x 1024
    (gfx950 tallies a 128-byte read request at 64 bytes) + WRITE_SIZE KiB x 1024; k_pose_reduce's known 42 B / pixel in the same passes
    is returned as the calibration.  Any failure (no rocprofv3, a timeout, an unexpected file layout) returns None and the line keeps
    the figure of profiles/pmc_traffic.json, labelled as such."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3')
    if exe is None:
        return None
    vals = {}
    try:
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = tempfile.mkdtemp(prefix='rpe_pmc_', dir='/tmp')
            try:                                          # (the scratch directory goes away on every path out of the pass)
                cmd = [exe, '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--', sys.executable, os.path.abspath(__file__),
                       '--steps', '1', '--warmup', '1', '--cpu-frames', '0', '--no-extras', '--batch', str(args.batch), '--height', str(args.height),
                       '--width', str(args.width), '--raft-iters', str(args.raft_iters), '--solver', args.solver, '--solver-iters', str(args.solver_iters)]
                if args.fp16_features:
                    cmd.append('--fp16-features')
                if args.conv_bf16x3:
                    cmd.append('--conv-bf16x3')
                r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=90)
                if r.returncode != 0:
                    return None
                acc = {'k_corr_lookup': [], 'k_pose_reduce': []}
                for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if row.get('Counter_Name') != ctr:
                            continue
                        for k in acc:
                            if k in row['Kernel_Name']:
                                acc[k].append(float(row['Counter_Value']))
            finally:
                shutil.rmtree(d, ignore_errors=True)
            if not acc['k_corr_lookup']:
                return None
            vals[ctr] = {k: (sum(v) / len(v), len(v)) for k, v in acc.items() if v}
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    rd = 2.0 * 1024.0 * vals['FETCH_SIZE']['k_corr_lookup'][0]
    wr = 1024.0 * vals['WRITE_SIZE']['k_corr_lookup'][0]
    out = {'traffic_bytes_per_launch': rd + wr, 'read_bytes': rd, 'write_bytes': wr, 'launches': vals['FETCH_SIZE']['k_corr_lookup'][1],
           'source': 'measured in this run: two child passes of bench.py (--steps 1 --warmup 1, same workload) under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE '
                     '--kernel-trace; read = 2 x FETCH_SIZE KiB x 1024 (gfx950 tallies 128-byte requests at 64), write = WRITE_SIZE KiB x 1024',
           'launches_averaged': 'EVERY k_corr_lookup launch of the child process -- its set-up pass, its warm-up step and its one step, all GRU iterations '
                                '(3 passes x raft_iters launches) -- not only the launches whose coordinates roofline.coordinates describes'}
    if 'k_pose_reduce' in vals['FETCH_SIZE']:
        out['calibration_pose_reduce_read_bytes'] = 2.0 * 1024.0 * vals['FETCH_SIZE']['k_pose_reduce'][0]
        # (round 6: ONE k_pose_reduce launch runs all evaluations of a solve; L-BFGS with N iterations = N evaluations)
        out['calibration_pose_reduce_algorithmic_bytes'] = args.batch * args.height * args.width * 42 * max(1, args.solver_iters)
        out['calibration_note'] = 'k_pose_reduce: one launch = the whole solve (solver_iters evaluations of 42 B / pixel)'
    return out


def lookup_rounds(pyr, coords):
    """Roughness of a set of lookup coordinates as the kernel sees it (rpe_corr_lookup_rounds, the kernel's own round rule):
    mean staging rounds per (level, group of 8 queries) -- 1.0 = every group's windows fit one box -- the fraction of groups that
    need more than one round, and the 128-B lines the loader requests relative to the single-round minimum of smooth flow."""
    rounds, lines = pyr.rounds(coords)
    r = rounds.float()
    return {'mean_rounds_per_group': float(r.mean()), 'multi_round_group_fraction': float((r > 1).float().mean()),
            'max_rounds': int(r.max()), 'loader_lines_requested': int(lines.sum())}


def lookup_realistic(pyr, out, B, H, W, dev):
    """The lookup kernel on coordinates a TRAINED network would produce: the ground-truth temporal flow (first B pairs) and stereo
    flow (last B pairs: disparities 8..60 px with depth edges) of seeded synthetic scenes with foreground occluders
    (synth.ground_truth_flows), sampled at the 1/8 grid's cell centres.  Same pyramid, same launch geometry, median of 60 launches timed by HIP events."""
    from rpe_amd import synth
    tf, sf = synth.ground_truth_flows(seed=777, n=B, h=H, w=W)
    flow8 = torch.cat((tf, sf))[:, :, 4::8, 4::8].contiguous() / 8.0
    h8, w8 = H // 8, W // 8
    ys, xs = torch.meshgrid(torch.arange(h8, dtype=torch.float32), torch.arange(w8, dtype=torch.float32), indexing='ij')
    coords = (torch.stack((xs, ys))[None] + flow8).contiguous().to(dev)
    # The scenes above are made on the host (seconds): the GPU has dropped to its idle clocks meanwhile, and a window of 30 launches (3 ms)
    # right after it once measured 313 us per launch.  ~40 ms of the same launches first, then the MEDIAN of 60 individually timed ones.
    for _ in range(400):
        pyr.lookup(coords, out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    for a, b in ev:
        a.record(); pyr.lookup(coords, out); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    t = ts[len(ts) // 2] / 1e3
    alg = lookup_algorithmic_bytes(2 * B, h8, w8)
    d = flow8[B:, 0]
    return {'kernel': 'k_corr_lookup', 'bound': 'hbm', 'achieved': alg / t / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': alg / t / 1e9 / HBM_PEAK_GBS, 'avg_launch_us': t * 1e6, 'algorithmic_bytes_per_launch': alg,
            'coordinates': f'ground-truth flow of synthetic scenes with 3 occluders per frame: {B} temporal + {B} stereo pairs, '
                           f'disparity {float(-d.max()) * 8:.0f}..{float(-d.min()) * 8:.0f} px',
            **lookup_rounds(pyr, coords)}


def deployment_numbers(args, model, cfg, frames, gpu_in, mask2_init, dev, solve_events, timing):
    """What the reference really deploys is batch 1, one frame after the other (scripts/infer_trajectory.py:57,71-77):
    ``latency_batch1_ms`` = median PoseNet.infer latency of ONE frame pair; ``tracker_fps`` = PoseEstimator over a
    24-frame sequence (L-BFGS 20 as configuration/infer_f2f.yaml:11, streaming reuse of the previous frame's encoder
    outputs); ``gn_ms_per_step`` / ``roofline_pose_solve_gn`` = the headline step with the Gauss-Newton solver mode."""
    import warnings
    from rpe_amd import pose_estimator, synth
    H, W, B = args.height, args.width, args.batch
    out = {}
    one = {k: v[:1].contiguous() for k, v in gpu_in.items()}
    m1 = mask2_init[:1].clone()
    lat = []
    for i in range(13):
        one['mask2'].copy_(m1)
        torch.cuda.synchronize()
        t = time.perf_counter()
        model.infer(**one)
        torch.cuda.synchronize()
        if i >= 3:
            lat.append(time.perf_counter() - t)
    out['latency_batch1_ms'] = 1e3 * sorted(lat)[len(lat) // 2]
    # sequential tracker
    F = 24
    seq = [(frames['image2l'][i % B:i % B + 1].to(dev), frames['image2r'][i % B:i % B + 1].to(dev), frames['mask2'][i % B:i % B + 1].to(dev))
           for i in range(F)]
    slam = dict(frame2frame=True, depth_clipping=[1, 250], lbgfs_iters=20, conf_weighing=True, reuse_features=True)
    keep = model.pose_head.problem.lbgfs_iters
    model.pose_head.problem.lbgfs_iters = 20
    try:
        for rep in range(2):
            est = pose_estimator.PoseEstimator(slam, frames['K'][0], 7.2 * 250.0, model, (W, H)).to(dev)
            torch.cuda.synchronize()
            t = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                for l, r, m in seq:
                    est(l, r, m.clone())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
    finally:
        model.pose_head.problem.lbgfs_iters = keep
    out['tracker_fps'] = F / dt
    out.update(tracker_split(model, seq, slam, frames, W, H, dev))
    # the same walk with the next frame's encoders prefetched on a side stream (PoseEstimator.submit / result; poses bit-identical)
    model.pose_head.problem.lbgfs_iters = 20
    try:
        for rep in range(2):
            est = pose_estimator.PoseEstimator(slam, frames['K'][0], 7.2 * 250.0, model, (W, H)).to(dev)
            torch.cuda.synchronize()
            t = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                est.submit(seq[0][0], seq[0][1], seq[0][2].clone())
                for i in range(F):
                    if i + 1 < F:
                        est.submit(seq[i + 1][0], seq[i + 1][1], seq[i + 1][2].clone())
                    est.result()
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t
    finally:
        model.pose_head.problem.lbgfs_iters = keep
    out['tracker_prefetch_fps'] = F / dtp
    out['tracker_prefetch_config'] = 'the same frames through PoseEstimator.submit / result: frame t+1 is encoded on a side stream while frame t\'s update loop runs (poses bit-identical)'
    # the same 24 frames in chunks of 16 (PoseEstimator.forward_chunk: one RAFT pass per chunk, bit-identical poses; what
    # SequenceTracker / bench.py --mode sequence run)
    L, R, Mk = (torch.cat([f[i] for f in seq]) for i in range(3))
    model.pose_head.problem.lbgfs_iters = 20
    try:
        for rep in range(2):
            est = pose_estimator.PoseEstimator(slam, frames['K'][0], 7.2 * 250.0, model, (W, H)).to(dev)
            torch.cuda.synchronize()
            t = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                est(L[:1], R[:1], Mk[:1].clone())
                for a in range(1, F, 16):
                    e = min(F, a + 16)
                    if e - a == 1:
                        est(L[a:e], R[a:e], Mk[a:e].clone())
                    else:
                        est.forward_chunk(L[a:e], R[a:e], Mk[a:e].clone())
            torch.cuda.synchronize()
            dtc = time.perf_counter() - t
    finally:
        model.pose_head.problem.lbgfs_iters = keep
    out['tracker_chunk16_fps'] = F / dtc
    out['tracker_config'] = f'PoseEstimator, {F} frames one at a time, {W}x{H}, 12 GRU iters, L-BFGS 20, encoder outputs of frame t reused at t+1'
    out['tracker_chunk16_config'] = 'the same frames through PoseEstimator.forward_chunk in chunks of 16 (one RAFT pass per chunk; poses bit-identical to one frame at a time)'
    # Gauss-Newton mode of the same step
    if args.solver != 'gn':
        prob = model.pose_head.problem
        prob.solver = 'gn'
        try:
            n0 = len(solve_events)

            def gstep():
                gpu_in['mask2'].copy_(mask2_init)
                timing['on'] = True
                try:
                    return model.infer(**gpu_in)
                finally:
                    timing['on'] = False
            gstep()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                gstep()
            torch.cuda.synchronize()
            out['gn_ms_per_step'] = 1e3 * (time.perf_counter() - t) / 3
            r = pose_roofline(solve_events[n0 + 1:], B, H, W, args.solver_iters)
            r['kernels'] = r['kernels'] + ' (Gauss-Newton: 21-entry Hessian per evaluation)'
            out['roofline_pose_solve_gn'] = r
        finally:
            prob.solver = 'lbfgs'
    return out


def tracker_split(model, seq, slam, frames, W, H, dev):
    """Is the sequential tracker host-bound or GPU-bound on THIS box?  The same frames as ``tracker_fps``, per steady-state frame (medians
    over frames 2..F): ``tracker_gpu_ms_per_frame`` = HIP events around PoseEstimator.forward; ``tracker_host_enqueue_ms_per_frame`` =
    perf_counter from the call to the moment the last launch has been handed to the runtime (PoseEstimator.t_enqueued, taken right
    before the success-flag synchronisation); ``tracker_wall_ms_per_frame``; the launch counts of one frame; and the same walk with
    the update loop launched call by call from Python (raft.LOOP_OPLIST = False) for the A/B on one box."""
    import statistics
    import warnings
    from rpe_amd import _lib as rl
    from rpe_amd import pose_estimator
    from rpe_amd import raft as raft_mod
    F = len(seq)
    res = {}

    def walk(est, rec=None):
        for i, (l, r, m) in enumerate(seq):
            mm = m.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            est(l, r, mm)
            e1.record()
            t1 = time.perf_counter()
            if rec is not None and i >= 2:
                rec.append((est.t_enqueued - t0, t1 - t0, e0, e1))
    keep = model.pose_head.problem.lbgfs_iters
    model.pose_head.problem.lbgfs_iters = 20
    keep_list, keep_frame = raft_mod.LOOP_OPLIST, raft_mod.FRAME_OPLISTS
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for oplist in (False, True):
                raft_mod.LOOP_OPLIST = raft_mod.FRAME_OPLISTS = oplist
                for rep in range(2):
                    rec = []
                    est = pose_estimator.PoseEstimator(slam, frames['K'][0], 7.2 * 250.0, model, (W, H)).to(dev)
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    walk(est, rec)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t
                med = lambda v: statistics.median(v)
                if oplist:
                    res['tracker_gpu_ms_per_frame'] = med([a.elapsed_time(b) for _, _, a, b in rec])
                    res['tracker_host_enqueue_ms_per_frame'] = 1e3 * med([h for h, _, _, _ in rec])
                    res['tracker_wall_ms_per_frame'] = 1e3 * med([w for _, w, _, _ in rec])
                    res['tracker_fps_with_events'] = F / dt
                else:
                    res['tracker_launch_by_launch'] = {'fps': F / dt, 'host_enqueue_ms_per_frame': 1e3 * med([h for h, _, _, _ in rec]),
                                                       'gpu_ms_per_frame': med([a.elapsed_time(b) for _, _, a, b in rec]),
                                                       'config': 'raft.LOOP_OPLIST = raft.FRAME_OPLISTS = False: every launch dispatched from Python (rounds 1-5)'}
            # launch counts of one steady-state frame
            raft_mod.LOOP_OPLIST = raft_mod.FRAME_OPLISTS = True
            est = pose_estimator.PoseEstimator(slam, frames['K'][0], 7.2 * 250.0, model, (W, H)).to(dev)
            for l, r, m in seq[:3]:
                est(l, r, m.clone())
            l, r, m = seq[3]
            with rl.CountingLib() as counter:
                est(l, r, m.clone())
            torch.cuda.synchronize()
            res['tracker_launches_per_frame'] = {'entry_point_calls_from_python': counter.calls, 'ops_enqueued_by_launch_lists': counter.list_ops,
                                                 'note': 'an entry point is one kernel launch except rpe_pose_solve_ex (2: k_pose_init + one persistent k_pose_reduce), rpe_unet_heads (15) and '
                                                         'rpe_corr_build_ex (2); kernel counts per frame: profiles/r06_tracker_kernel_stats_last_frame.txt'}
    finally:
        model.pose_head.problem.lbgfs_iters = keep
        raft_mod.LOOP_OPLIST, raft_mod.FRAME_OPLISTS = keep_list, keep_frame
    res['tracker_bound'] = 'gpu' if res['tracker_host_enqueue_ms_per_frame'] <= 0.5 * res['tracker_gpu_ms_per_frame'] else 'host (enqueue > half of the GPU time)'
    return res


def pose_roofline(events, frames, h, w, iters):
    """Second HBM-bound kernel family: the whole device-resident solve (N x (k_pose_reduce + k_pose_update)), HIP-event
    timed in the timed region; algorithmic bytes = 42 B/pixel per evaluation."""
    if not events:
        return None
    t = sum(a.elapsed_time(b) for a, b in events) / len(events) / 1e3
    alg = frames * h * w * 42 * iters
    return {'kernels': 'k_pose_init + k_pose_reduce x%d (the L-BFGS / GN update runs in the tail of each reduction)' % iters, 'bound': 'hbm', 'achieved': alg / t / 1e9, 'peak': HBM_PEAK_GBS,
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
    cp = torch.cat(poses).reshape(-1, 7)
    diff = float((cp - gp).abs().max())
    # BASELINE.json's "ATE-RMSE vs ref": both sets of relative poses chained into trajectories the way the tracker does
    # (pose_estimator.py:90-91, x250 de-normalised, i.e. millimetres) and compared with the reference's own metric definitions
    from oracle import tracker as otracker
    from rpe_amd import trajectory
    Tc, Tg = (trajectory.pose_matrices(otracker.chain(x.float(), 250.0).numpy()) for x in (cp, gp))
    ate, _ = trajectory.absolute_trajectory_error(Tc, Tg, prealign=False)
    rpe_t, rpe_r = trajectory.relative_pose_error(Tc, Tg)
    return {'value': n_frames / dt, 'unit': 'pose solves/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n_frames} frame pairs of the same batch, one at a time, torch CPU f32 + f64 L-BFGS',
            'max_abs_pose_diff_vs_gpu': diff, 'ate_rmse_gpu_vs_cpu_mm': ate, 'rpe_trans_gpu_vs_cpu_mm': float(rpe_t.mean()),
            'rpe_rot_gpu_vs_cpu_rad': float(rpe_r.mean())}


if __name__ == '__main__':
    main()
