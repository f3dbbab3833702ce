#!/usr/bin/env python3
"""A/B helper for the register-chained first-level tile: time the POS model alone (its first level is body-part
tiles only - no 32-row trajectory tiles between them), the TRJ model alone and the pair, in one process, for the tile the
environment selects (hooks build: R3D_USE_HOOKS_LIB=1 R3D_CHAIN=1|0).  usage: python tools/chain_pos_only.py [windows] [steps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ray3d_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
lifter, states = bench.build(dev)
cp = states["pos"][0]
x = torch.from_numpy(synth.synth_rays(B, cp, seed=5)).to(dev)
p = torch.from_numpy(synth.synth_param(B, seed=6)).to(dev)


def timed(fn):
    with torch.no_grad():
        for _ in range(40):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        e1.synchronize()
    return e0.elapsed_time(e1) / steps


res = {"pos": timed(lambda: lifter.pos(x, p)), "trj": timed(lambda: lifter.trj(x, p)), "pair": timed(lambda: lifter(x, p))}
with torch.no_grad():
    recs = lifter.profile_call(lambda: lifter.pos(x, p), dev)
lifter.check_status()
print("CHAIN=%s B=%d  pos alone %.4f ms  trj alone %.4f ms  pair %.4f ms  kernels(pos): %s" % (
    os.environ.get("R3D_CHAIN", "-"), B, res["pos"], res["trj"], res["pair"], sorted(set(r["kernel"] for r in recs))))
