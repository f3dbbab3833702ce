#!/usr/bin/env python3
"""Error of the fp32-MFMA path and of the bf16x3 path against a FLOAT64 evaluation of the same network (oracle/torch_port.py
in double precision), over several weight / input seeds and output scales.  Development evidence for DESIGN.md section 4.4;
the pass/fail version of it is tests/test_gpu_parity.py::test_bf16x3_error_against_float64_is_the_fp32_paths.
usage (on a GPU box): python tools/bf16x3_error_table.py [seeds] [windows]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray3d_amd
from ray3d_amd import synth
from ray3d_amd.spec import config_from_dicts
from oracle import torch_port

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.set_num_threads(min(64, os.cpu_count() or 8))
print("| seed | output scale | max abs output | fp32 MFMA: max / mean abs error | bf16x3: max / mean abs error | fp32 vs bf16x3 max diff |")
print("|---|---|---|---|---|---|")
for seed in range(nseeds):
    scale = (1.0, 1.0, 8.0, 0.25, 1.0, 30.0)[seed % 6]
    mc = ray3d_amd.default_model_config(ARCHITECTURE="3,3,3,3,3")
    cp, ct = config_from_dicts(mc, "pos"), config_from_dicts(mc, "trj")
    sp, st = synth.synth_state(cp, seed=11 + 2 * seed, out_scale=scale), synth.synth_state(ct, seed=12 + 2 * seed, out_scale=scale)
    x, p = synth.synth_rays(B, cp, seed=100 + seed), synth.synth_param(B, seed=200 + seed)
    sd64 = [{k: torch.from_numpy(np.asarray(v)).double() for k, v in s_.items() if np.asarray(v).dtype == np.float32} for s_ in (sp, st)]
    with torch.no_grad():
        x64, p64 = torch.from_numpy(x).double(), torch.from_numpy(p).double()
        ref = (torch_port.forward(cp, sd64[0], x64, p64) + torch_port.forward(ct, sd64[1], x64, p64)).numpy()
    outs = {}
    for mode in (False, True):
        fac = ray3d_amd.Model(dict(mc, BF16X3=mode), {}, is_train=False)
        pos, trj = fac.get_pos_model(), fac.get_trj_model()
        pos.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sp.items()}, strict=True)
        trj.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, strict=True)
        with torch.no_grad():
            outs[mode] = ray3d_amd.Ray3DLifter(pos.cuda(), trj.cuda()).eval()(torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()).cpu().numpy().astype(np.float64)
    e32, e3 = np.abs(outs[False] - ref), np.abs(outs[True] - ref)
    print("| %d | %g | %.2f | %.2e / %.2e | %.2e / %.2e | %.2e |" % (seed, scale, np.abs(ref).max(), e32.max(), e32.mean(), e3.max(), e3.mean(),
                                                                     np.abs(outs[False] - outs[True]).max()))
