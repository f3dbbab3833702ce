"""Development aid: every window of calls at the window counts where the plan (and the bf16x3 threshold) switches, both
precisions, against the torch port of the reference graph."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import test_gpu_parity as T
import ray3d_amd
from ray3d_amd import synth
for arch, sizes in (("3,3,3,3,3", [47, 48, 49, 95, 96, 97, 1023, 1024, 1025]), ("3,3,3", [48, 49, 96, 97, 130, 1024, 2000])):
    for b3 in (False, True):
        mc = ray3d_amd.default_model_config(ARCHITECTURE=arch, BF16X3=b3)
        pos, trj, (cp, sp), (ct, st) = T.build_modules(mc)
        lifter = ray3d_amd.Ray3DLifter(pos, trj).eval()
        for B in sizes:
            x, p = synth.synth_rays(B, cp, seed=B), synth.synth_param(B, seed=B + 1)
            with torch.no_grad():
                out = lifter(torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()).cpu().numpy()
            ref = T._oracle_lift(((cp, sp), (ct, st)), x, p)
            err = np.abs(out - ref).max()
            assert err <= T.tol_for(ref), (arch, b3, B, err)
            print(arch, "b3" if b3 else "f32", B, "%.2e" % err)
print("all plan-switch sizes ok")
