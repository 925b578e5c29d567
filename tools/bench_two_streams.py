"""Experiment: the headline step (16 frame pairs) as ONE batch-16 infer vs TWO batch-8 infers on two HIP streams (the tail of one
stream's launch filled by the other's).  python tools/bench_two_streams.py [--steps 10]"""
import argparse
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--ways', type=int, default=2)
    args = ap.parse_args()
    import rpe_amd  # noqa: F401
    from rpe_amd import pose_net, synth
    dev = torch.device('cuda:0')
    H, W, B = 512, 640, args.batch
    cfg = synth.model_config(H, W, iters=12, lbgfs_iters=8)
    model = pose_net.PoseNet(cfg)
    synth.init_synthetic_weights(model, seed=1234)
    model.eval().to(dev)
    frames = synth.stereo_frames(seed=1000, n=B, h=H, w=W)
    full = {k: v.to(dev) for k, v in synth.infer_args(frames).items()}

    def run(models, inputs, streams, steps):
        masks = [i['mask2'].clone() for i in inputs]

        def step():
            outs = []
            for m, i, s, mk in zip(models, inputs, streams, masks):
                with torch.cuda.stream(s):
                    i['mask2'].copy_(mk)
                    outs.append(m.infer(**i, ret_details=False))
            return outs
        for _ in range(2):
            out = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, out

    with torch.no_grad():
        t1, o1 = run([model], [full], [torch.cuda.current_stream()], args.steps)
        print(f'one batch-{B} infer          : {t1:8.2f} ms/step  {B / t1 * 1e3:7.1f} solves/s')
        n = args.ways
        models = [model] + [copy.deepcopy(model) for _ in range(n - 1)]
        per = B // n
        parts = [{k: (v[j * per:(j + 1) * per].contiguous() if v.shape[0] == B else v) for k, v in full.items()} for j in range(n)]
        streams = [torch.cuda.Stream() for _ in range(n)]
        t2, o2 = run(models, parts, streams, args.steps)
        print(f'{n} batch-{per} infers on {n} streams: {t2:8.2f} ms/step  {B / t2 * 1e3:7.1f} solves/s')
        def vec(o):
            o = o[0] if isinstance(o, (tuple, list)) else o
            return o.vec() if hasattr(o, 'vec') else o
        print('max |pose diff| between the two schedules:', float((vec(o1[0]) - torch.cat([vec(o) for o in o2])).abs().max()))


if __name__ == '__main__':
    main()
