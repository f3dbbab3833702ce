"""Development aid: host time of r3d_set_weight x 561 + r3d_finalize for the RF-243 pos + trj pair, with and without the shrink fold."""
import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ray3d_amd
from ray3d_amd import synth
from ray3d_amd.spec import config_from_dicts
for nofold in ("0", "1"):
    os.environ["R3D_NO_SHRINK_FOLD"] = nofold
    mc = ray3d_amd.default_model_config(ARCHITECTURE="3,3,3,3,3")
    fac = ray3d_amd.Model(mc, {}, is_train=False)
    pos, trj = fac.get_pos_model(), fac.get_trj_model()
    for m, kind, seed in ((pos, "pos", 1), (trj, "trj", 2)):
        cfg = config_from_dicts(mc, kind)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state(cfg, seed=seed).items()}, strict=True)
    lifter = ray3d_amd.Ray3DLifter(pos.cuda(), trj.cuda()).eval()
    torch.cuda.synchronize()
    t = time.perf_counter()
    lifter.pos.handle(torch.device("cuda:0")); lifter.trj.handle(torch.device("cuda:0"))
    torch.cuda.synchronize()
    print("no_fold=%s: set_weight + finalize of pos and trj: %.0f ms" % (nofold, (time.perf_counter() - t) * 1e3))
