#!/usr/bin/env python3
"""stage_times.py for the pos network ALONE (no trajectory tiles beside the body-part ones in the first-level launch)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray3d_amd
from ray3d_amd import synth, _capi
if os.environ.get("R3D_LIB_OVERRIDE"):
    _capi.LIB_PATH = os.path.abspath(os.environ["R3D_LIB_OVERRIDE"])
from ray3d_amd.spec import config_from_dicts
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mc = ray3d_amd.default_model_config(ARCHITECTURE="3,3,3,3,3")
fac = ray3d_amd.Model(mc, {}, is_train=False)
pos = fac.get_pos_model()
cfg = config_from_dicts(mc, "pos")
pos.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state(cfg, seed=1).items()}, strict=True)
pos = pos.cuda().eval()
x = torch.from_numpy(synth.synth_rays(B, cfg, seed=3)).cuda()
p = torch.from_numpy(synth.synth_param(B, seed=4)).cuda()
with torch.no_grad():
    for _ in range(5):
        out = pos(x, p)
torch.cuda.synchronize()
print("done", float(out.abs().max()))
