"""Development aid: time per step of the first 400 steps after two idle seconds (the clock ramp bench.py's settle_clocks waits out)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
dev = torch.device("cuda:0")
lifter, states = bench.build(dev)
from ray3d_amd import synth
cfg = states["pos"][0]
x = torch.from_numpy(synth.synth_rays(256, cfg, seed=100)).to(dev)
p = torch.from_numpy(synth.synth_param(256, seed=0, vary=False)).to(dev)
with torch.no_grad():
    lifter.prepare([256], dev)
    lifter(x, p)
    torch.cuda.synchronize()
    time.sleep(2.0)                       # let the clocks fall back
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(402)]
    ev[0].record()
    for i in range(401):
        lifter(x, p)
        ev[i + 1].record()
    torch.cuda.synchronize()
t = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(401)])
for a, b in ((0, 5), (5, 25), (25, 50), (50, 100), (100, 200), (200, 400)):
    print("steps %3d-%3d: %.4f ms/step" % (a, b, t[a:b].mean()))
