#!/usr/bin/env python3
"""Per-launch times of one forward (library's own HIP events), median of a few runs.
usage: [ARCH=3,3 J=14] stage_times.py [B] [reps]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray3d_amd
from ray3d_amd import synth, _capi
if os.environ.get("R3D_LIB_OVERRIDE"):          # A/B against another build of the library
    _capi.LIB_PATH = os.path.abspath(os.environ["R3D_LIB_OVERRIDE"])
from ray3d_amd.spec import config_from_dicts

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
mc = ray3d_amd.default_model_config(ARCHITECTURE=os.environ.get("ARCH", "3,3,3,3,3"), NUM_KPTS=int(os.environ.get("J", "17")))
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
    for _ in range(3):
        lifter.forward(x, p)
    runs = [lifter.profile(x, p) for _ in range(reps)]
ms = np.median(np.array([[r["ms"] for r in run] for run in runs]), axis=0)
tot = 0.0
for r, t in zip(runs[0], ms):
    tot += t
    print("%2d %-18s wgs %4d  %7.1f us  %6.1f TF" % (r["stage"], r["kernel"], r["blocks"], t * 1e3, r["flops"] / t / 1e9 if t > 0 else 0))
print("sum %.1f us" % (tot * 1e3))
