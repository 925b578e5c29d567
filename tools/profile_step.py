#!/usr/bin/env python3
"""Phase timing of one bench step (HIP events around the phases of PoseNet.infer), to see where a step's time goes.
Options let us try conv-library settings: --benchmark (MIOpen exhaustive find), --channels-last."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops, pose_net, synth, raft as raft_mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--benchmark', action='store_true')
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    if a.benchmark:
        torch.backends.cudnn.benchmark = True
    dev = torch.device('cuda:0')
    H, W, B = 512, 640, a.batch
    cfg = synth.model_config(H, W)
    model = synth.init_synthetic_weights(pose_net.PoseNet(cfg)).eval().to(dev)
    fr = synth.stereo_frames(1000, B, H, W)
    g = {k: v.to(dev) for k, v in synth.infer_args(fr).items()}
    m2 = g['mask2'].clone()
    marks = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))

    R = model.flow
    ub = R.update_block

    @torch.no_grad()
    def step():
        marks.clear()
        g['mask2'].copy_(m2)
        n = B
        mark('start')
        i1 = torch.cat((g['image1l'], g['image2l'])); i2 = torch.cat((g['image2l'], g['image2r']))
        N = 2 * n; h8, w8 = H // 8, W // 8
        im1 = 2 * (i1 / 255.0) - 1.0; im2 = 2 * (i2 / 255.0) - 1.0
        f = R.fnet(torch.cat((im1, im2)))
        mark('fnet')
        cnet = R.cnet(im1)
        mark('cnet')
        pyr = R._pyramid(N, h8, w8, dev).build(f[:N].float(), f[N:].float())
        mark('corr_build')
        c = 128
        hx = torch.empty(N, 2 * c, h8, w8, device=dev); rhx = torch.empty_like(hx); z = torch.empty(N, c, h8, w8, device=dev)
        hb = torch.empty(N, c, h8, w8, device=dev); catb = torch.empty(N, 2 * c, h8, w8, device=dev); F = torch.nn.functional
        torch.tanh(cnet[:, :c], out=hx[:, :c]); inp = torch.relu(cnet[:, c:]); ctx = ub.context_terms(inp); GW = ub.gate_weights()
        c0 = raft_mod.coords_grid(N, h8, w8, dev); c1 = c0.clone()
        corr = torch.empty(N, 324, h8, w8, device=dev)
        mark('init')
        tl = tm = tg = tf = 0
        for it in range(12):
            pyr.lookup(c1, out=corr); mark('lookup')
            flow = c1 - c0
            ub.encoder(flow, corr, catb, hx, rhx); mark('motion_enc')
            zr = F.conv2d(hx, GW['zr1'][0], None, padding=(0, 2)); ops.gru_gates_zr(zr, hx, c, z, rhx, add=ctx['zr1'])
            q = F.conv2d(rhx, GW['q1'][0], None, padding=(0, 2)); ops.gru_gates_h(z, q, hx, c, hx, add=ctx['q1'])
            zr = F.conv2d(hx, GW['zr2'][0], None, padding=(2, 0)); ops.gru_gates_zr(zr, hx, c, z, rhx, add=ctx['zr2'])
            q = F.conv2d(rhx, GW['q2'][0], None, padding=(2, 0)); ops.gru_gates_h(z, q, hx, c, hx, add=ctx['q2']); mark('gru')
            hb.copy_(hx[:, :c]); fh = ub.flow_head
            c1 = ops.conv3x3_to2(ops.bias_act(F.conv2d(hb, fh.conv1.weight, None, padding=1), fh.conv1.bias), fh.conv2.weight, fh.conv2.bias, add=c1); mark('flow_head')
        up = ops.upsample_convex(c1 - c0, ub.up_mask(hb)); mark('mask+upsample')
        tfl = up[:n].contiguous(); sf2 = up[n:].contiguous()
        gg = ops.depth_backproject_warp(sf2, tfl, g['baseline'], g['intrinsics'], g['depth1'], g['image1l'], g['image2l'], g['stereo_flow1'], g['mask2'])
        mark('geometry')
        hid = hb[:n]; ctx = inp[:n]
        w2d = model.weight_head_2d(torch.cat((gg['inp1'], hid, ctx), 1)); w3d = model.weight_head_3d(torch.cat((gg['inp1'], gg['inp2'], hid, ctx), 1))
        mark('weight_heads')
        lw = model.loss_weight.detach()[None].repeat(n, 1)
        ops.pose_solve(tfl, gg['pcl1'], gg['pcl2w'], w2d, w3d, g['mask1'], gg['mask2w'], g['intrinsics'], lw, iters=8)
        mark('solve')

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    tot = {}
    t0 = time.perf_counter()
    for _ in range(a.reps):
        step(); torch.cuda.synchronize()
        prev = marks[0][1]
        for name, e in marks[1:]:
            tot[name] = tot.get(name, 0.0) + prev.elapsed_time(e); prev = e
    wall = (time.perf_counter() - t0) / a.reps * 1e3
    s = sum(tot.values()) / a.reps
    print(f'batch {B} frames: wall {wall:.1f} ms/step, event sum {s:.1f} ms  benchmark={a.benchmark}')
    for k, v in tot.items():
        print(f'  {k:14s} {v / a.reps:8.2f} ms  {100 * v / a.reps / s:5.1f}%')


if __name__ == '__main__':
    main()
