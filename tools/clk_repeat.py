#!/usr/bin/env python3
"""The live clock reading (r3d_last_clock) of consecutive forwards next to their duration: usage clk_repeat.py [B] [n]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
from ray3d_amd import synth
dev = torch.device("cuda:0")
lifter, states = bench.build(dev)
cfg = states["pos"][0]
x = torch.from_numpy(synth.synth_rays(B, cfg, seed=100)).to(dev)
p = torch.from_numpy(synth.synth_param(B, seed=0, vary=False)).to(dev)
with torch.no_grad():
    lifter.prepare([B], dev)
    for _ in range(100):
        lifter(x, p)
    rows = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(20):
            lifter(x, p)
        e0.record(); lifter(x, p); e1.record(); e1.synchronize()
        rows.append((e0.elapsed_time(e1) * 1e3, lifter.last_clock_ghz(dev)))
us = np.array([r[0] for r in rows]); ck = np.array([r[1] for r in rows])
print("B %d: forward+decoder %.1f us (min %.1f max %.1f); clk median %.3f min %.3f max %.3f GHz; us x GHz = %.0f k cycles (min %.0f max %.0f)" % (
    B, np.median(us), us.min(), us.max(), np.median(ck), ck.min(), ck.max(), np.median(us * ck), (us * ck).min(), (us * ck).max()))
