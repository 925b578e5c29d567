import os, sys, time, torch
sys.path.insert(0, os.getcwd())
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ['/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us']:
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
print('loadavg', open('/proc/loadavg').read())
import rpe_amd
from rpe_amd import synth
from oracle import pose_net as opn
H,W=352,384
cfg = synth.model_config(H,W)
om = opn.PoseNet(cfg).eval()
fr = synth.stereo_frames(0,1,H,W); a = synth.infer_args(fr)
for nt in [8, 16, 32, 64, 128]:
    torch.set_num_threads(nt)
    t=time.time(); om.stages(**{k:v.clone() for k,v in a.items()}); dt=time.time()-t
    t=time.time(); om.stages(**{k:v.clone() for k,v in a.items()}); dt2=time.time()-t
    print(nt, 'threads: stages', round(dt,2), round(dt2,2), flush=True)
