import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpe_amd
from rpe_amd import pose_net, synth
dev = torch.device('cuda:0')
H, W = 512, 640
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = synth.model_config(H, W)
model = synth.init_synthetic_weights(pose_net.PoseNet(cfg)).eval().to(dev)
fr = synth.stereo_frames(1000, B, H, W)
g = {k: v.to(dev) for k, v in synth.infer_args(fr).items()}
m2 = g['mask2'].clone()
def step():
    g['mask2'].copy_(m2)
    return model.infer(**g, ret_details=True)
for _ in range(3): out = step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): out = step()
torch.cuda.synchronize(); eager = (time.perf_counter() - t) / 10 * 1e3
ref = out[0].data.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    gout = step()
torch.cuda.synchronize()
for _ in range(3): gr.replay()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): gr.replay()
torch.cuda.synchronize(); graphed = (time.perf_counter() - t) / 10 * 1e3
print(f'batch {B}: eager {eager:.2f} ms/step, graph replay {graphed:.2f} ms/step; pose diff {float((gout[0].data - ref).abs().max()):.2e}')
