import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from ray3d_amd import synth
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
def run(nsplit, B=256, steps=50, warm=10):
    lifters = []
    for i in range(nsplit):
        l, states = bench.build(dev)
        lifters.append(l)
    cfg = states["pos"][0]
    xs = [torch.from_numpy(synth.synth_rays(B // nsplit, cfg, seed=100 + i)).to(dev) for i in range(nsplit)]
    ps = [torch.from_numpy(synth.synth_param(B // nsplit, seed=0, vary=False)).to(dev) for i in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    def step():
        for l, x, p, s in zip(lifters, xs, ps, streams):
            with torch.cuda.stream(s):
                l(x, p)
    with torch.no_grad():
        for _ in range(warm): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("split %d x %d: %.4f ms/step -> %.0f poses/s" % (nsplit, B // nsplit, dt / steps * 1e3, B * steps / dt))
for n in (1, 2, 4):
    run(n)
run(2, B=512)
run(1, B=512)
