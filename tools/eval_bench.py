#!/usr/bin/env python3
"""Clip-sharded evaluation throughput (BASELINE configs[2] stand-in): synthetic clips with the Human3.6M evaluation
shape - lengths ~ U(1000, 6000) frames, four cameras - lifted with in-kernel sliding windows (forward_clip), errors
summed on the device (r3d_clip_metrics), per-clip partial rows gathered once.  usage: eval_bench.py [n_clips] [flip 0/1]
Under torch.distributed.run every rank evaluates its share of the clips (RCCL all_gather of the partial rows)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray3d_amd
from ray3d_amd import synth, evaluate
from ray3d_amd.spec import config_from_dicts

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 60
flip = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
group = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
mc = ray3d_amd.default_model_config(ARCHITECTURE="3,3,3,3,3")
fac = ray3d_amd.Model(mc, {}, is_train=False)
pos, trj = fac.get_pos_model(), fac.get_trj_model()
for m, kind, seed in ((pos, "pos", 1), (trj, "trj", 2)):
    cfg = config_from_dicts(mc, kind)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state(cfg, seed=seed).items()}, strict=True)
lifter = ray3d_amd.Ray3DLifter(pos.to(dev), trj.to(dev)).eval()
rng = np.random.default_rng(0)
cams = [ray3d_amd.synthetic_camera(yaw, 4.5, -12.0, name="cam%d" % i) for i, yaw in enumerate((20, 110, 200, 290))]
clips = []
for i in range(n_clips):
    n = int(rng.integers(1000, 6001))
    cam = cams[i % 4]
    world_pts = rng.normal(0, 0.3, (n, 17, 3)) + np.array([0, 0, 1.0])
    rays = cam.rays_from_uv(cam.project(world_pts)).astype(np.float32)
    clips.append(evaluate.Clip(cam, rays, cam.world2normalized(world_pts).astype(np.float32), "A%d" % (i % 15), i))
frames = sum(c.rays.shape[0] for c in clips)
kl, kr = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
with torch.no_grad():
    for rep in range(2):            # the second pass has every tile schedule cached
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        named, avg, rows = evaluate.evaluate_clips(lifter.forward_clip, clips, 243, dev, flip=flip, kps_left=kl, kps_right=kr,
                                                   rank=rank, world_size=world, group=group)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rank == 0:
            print("pass %d: %d clips, %d frames, %d GPU(s), flip=%d: %.3f s -> %.0f poses/s (MPJPE avg %.1f mm)"
                  % (rep, n_clips, frames, world, flip, dt, frames / dt, avg[0]))
