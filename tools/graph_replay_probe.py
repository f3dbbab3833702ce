#!/usr/bin/env python3
"""Does replaying the forward's 14 launches as a captured HIP graph beat enqueueing them?  usage: graph_test.py [B]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray3d_amd
from ray3d_amd import synth
from ray3d_amd.spec import config_from_dicts

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mc = ray3d_amd.default_model_config(ARCHITECTURE="3,3,3,3,3")
fac = ray3d_amd.Model(mc, {}, is_train=False)
pos, trj = fac.get_pos_model(), fac.get_trj_model()
for m, kind, seed in ((pos, "pos", 1), (trj, "trj", 2)):
    cfg = config_from_dicts(mc, kind)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state(cfg, seed=seed).items()}, strict=True)
lifter = ray3d_amd.Ray3DLifter(pos.cuda(), trj.cuda()).eval()
cp = config_from_dicts(mc, "pos")
x = torch.from_numpy(synth.synth_rays(B, cp, seed=3)).cuda()
p = torch.from_numpy(synth.synth_param(B, seed=4)).cuda()
with torch.no_grad():
    for _ in range(5):
        ref = lifter(x, p)
    torch.cuda.synchronize()
    def timeit(fn, n=100):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print("stream launches: %.4f ms" % timeit(lambda: lifter(x, p)))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        lifter(x, p)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = lifter(x, p)
    g.replay(); torch.cuda.synchronize()
    print("graph replay:    %.4f ms   equal: %s" % (timeit(g.replay), torch.equal(out, ref)))
    print("stream launches: %.4f ms" % timeit(lambda: lifter(x, p)))
