#!/usr/bin/env python3
"""Where does a row computed in a batch first differ from the same row computed alone?  (bisecting helper for the chunked tracker)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import ops, pose_net, synth, raft

H, W = int(os.environ.get('DH', 352)), int(os.environ.get('DW', 384))
cfg = synth.model_config(H, W, iters=12, lbgfs_iters=8)
model = synth.init_synthetic_weights(pose_net.PoseNet(cfg)).eval().cuda()
fr = synth.stereo_frames(21, 6, H, W)
L, R = fr['image2l'].cuda(), fr['image2r'].cuda()


def cmp(name, a, b):
    d = (a.float() - b.float()).abs().max().item()
    print(f'{name:28s} equal={torch.equal(a, b)}  maxdiff={d:.3e}')


with torch.no_grad():
    f_all = model.flow.encode_features((L, R))
    c_all = model.flow.encode_context(L)
    for i in (0, 3):
        cmp(f'fnet row {i} (batch 12 vs 2)', model.flow.encode_features((L[i:i + 1], R[i:i + 1]))[:1], f_all[i:i + 1])
        cmp(f'cnet row {i} (batch 6 vs 1)', model.flow.encode_context(L[i:i + 1]), c_all[i:i + 1])
    # encoder layer by layer
    enc = model.flow.fnet
    x6, n6 = enc._stem_many([L])
    x1, n1 = enc._stem_many([L[2:3]])
    cmp('stem raw', x1, x6[2:3])
    if n6 is not None:
        cmp('stem norm', n1, n6[2:3])
    a6 = enc.layer1[0](x6, n6); a1 = enc.layer1[0](x1, n1)
    cmp('layer1[0]', a1, a6[2:3])
    a6 = enc.layer1[1](a6); a1 = enc.layer1[1](a1)
    cmp('layer1[1]', a1, a6[2:3])
    b6 = enc.layer2[0](a6); b1 = enc.layer2[0](a1)
    cmp('layer2[0]', b1, b6[2:3])
    b6 = enc.layer2[1](b6); b1 = enc.layer2[1](b1)
    cmp('layer2[1]', b1, b6[2:3])
    c6 = enc.layer3(b6); c1 = enc.layer3(b1)
    cmp('layer3', c1, c6[2:3])
    cmp('final', enc._final(c1, False), enc._final(c6, False)[2:3])
    # RAFT on given features
    fm1, fm2 = f_all[:6], f_all[6:]
    flows6, hid6, ctx6 = model.flow(None, None, fmaps=(fm1, fm2), cnet=c_all)
    for sl in (slice(0, 1), slice(2, 4)):
        flows, hid, ctx = model.flow(None, None, fmaps=(fm1[sl].contiguous(), fm2[sl].contiguous()), cnet=c_all[sl].contiguous())
        cmp(f'raft flow rows {sl}', flows[-1], flows6[-1][sl]); cmp('raft hidden', hid, hid6[sl])
    # correlation / lookup alone
    pyr6 = ops.CorrPyramid(6, H // 8, W // 8).build(fm1, fm2)
    pyr1 = ops.CorrPyramid(1, H // 8, W // 8).build(fm1[2:3].contiguous(), fm2[2:3].contiguous())
    co = raft.coords_grid(6, H // 8, W // 8, 'cuda') + 0.37
    cmp('lookup', pyr1.lookup(co[2:3].contiguous()), pyr6.lookup(co)[2:3])
    # geometry + heads + solve
    a = synth.infer_args(synth.stereo_frames(5, 6, H, W))
    g = {k: v.cuda() for k, v in a.items()}
    s6 = model.stages(**{k: v.clone() for k, v in g.items()})
    s1 = model.stages(**{k: v[2:3].clone() for k, v in g.items()})
    for k in ('time_flow', 'depth2', 'pcl1', 'pcl2w', 'mask2w', 'w2d', 'w3d', 'inp1', 'inp2'):
        cmp('stages ' + k, s1[k], s6[k][2:3])
    lw = torch.ones(6, 2, device='cuda')
    args6 = (s6['time_flow'], s6['pcl1'], s6['pcl2w'], s6['w2d'], s6['w3d'], g['mask1'], s6['mask2w'], s6['intrinsics'], lw)
    T6 = ops.pose_solve(*args6, iters=8, partition_rows=1)[0]
    T1 = ops.pose_solve(*[x[2:3].contiguous() for x in args6], iters=8)[0]
    cmp('solve (partition_rows=1)', T1, T6[2:3])
    d6, v6 = ops.flow2depth(s6['stereo_flow2'], g['baseline'])
    cmp('flow2depth vs geom depth2', d6, s6['depth2'])
