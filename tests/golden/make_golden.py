#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by IMPORTING the reference on CPU.

Run in the build container only (needs /root/reference; the GPU box never has it):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

What is pinned:
  1. model_<case>.npz : for several (J, F, RF, stage, camera-embedding) configurations, the
     reference RIEModel / RIETrajectoryModel outputs and intermediate taps for the deterministic
     synthetic weights/inputs of ray3d_amd.synth (loaded with load_state_dict(strict=True), which
     also pins our state_dict key grammar and shapes against the reference constructors).
  2. cameras.npz      : CameraInfoPacket-derived constants and uv -> ray pairs for real camera
     calibrations taken from the reference's H36M / 3DHP tables (numbers only).
  3. evalcore.npz     : Trainer.evaluate_core's five metrics (with and without flip TTA) and its
     return_predictions output on synthetic clips.
  4. losses.npz       : mpjpe / n_mpjpe / p_mpjpe / mean_velocity_error known answers.
  6. frontends.npz    : the camera-augmented H36M front end (JSON camera list, scaled subjects, per-camera fetches) and
     HumanEva-I's camera layout / joint conventions.
  7. cameras_3dhp.npz : the 14 MPI-INF-3DHP cameras (S1/Seq1) as CameraInfoPacket builds them, with uv -> ray pairs.
  8. undistort.npz    : the reference's forward distortion model (data/camera_augmentation.py:502-542) on a pixel grid
     for the four H36M coefficient sets - the known-answer side of the (otherwise cv2-only) undistortion.
  5. dataset.npz      : the reference's Data front end on a small synthetic archive pair: per-clip ground truth in
     the normalised frame and ray-encoded keypoints, the left/right index lists, the H36M joint selection.
Only numbers leave this script; no reference source text is stored.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("RAY3D_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))   # only dereferenced on paths we do not take

from lib.model import Model as RefModel                     # noqa: E402
from lib.camera.camera import CameraInfoPacket              # noqa: E402
from lib.dataloader.generators import UnchunkedGenerator    # noqa: E402
from lib.train_val.trainer import Trainer                   # noqa: E402
from lib.loss import loss as ref_loss                        # noqa: E402
from lib.dataset.h36m_dataset import (h36m_cameras_extrinsic_params,      # noqa: E402
                                      h36m_cameras_intrinsic_params)
from lib.dataset.mpii_3dhp_dataset import camera_params as dhp_camera_params  # noqa: E402

from ray3d_amd.spec import config_from_dicts, default_model_config   # noqa: E402
from ray3d_amd import synth                                          # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(4)

CASES = {
    # name: (model_config overrides, batch, out_scale)
    "j17_rf27_s3": (dict(ARCHITECTURE="3,3,3"), 3, 1.0),
    "j17_rf243_s3": (dict(ARCHITECTURE="3,3,3,3,3"), 2, 1.0),
    "j17_rf9_s1": (dict(ARCHITECTURE="3,3", STAGE=1), 3, 1.0),
    "j14_rf9_s3": (dict(ARCHITECTURE="3,3", NUM_KPTS=14), 4, 1.0),
    "j15_rf9_s3": (dict(ARCHITECTURE="3,3", NUM_KPTS=15), 3, 1.0),
    "j17_f2_rf27_noemb_s3": (dict(ARCHITECTURE="3,3,3", INPUT_DIM=2, CAMERA_EMBDDING=False), 3, 1.0),
    "j17_rf81_s2_big": (dict(ARCHITECTURE="3,3,3,3", STAGE=2), 2, 8.0),
    # the dilated (non-Optimize1f) convolutions on RF-long windows, plain and causal (rie.py:91-92)
    "j17_rf27_dilated_s3": (dict(ARCHITECTURE="3,3,3", DISABLE_OPTIMIZATIONS=True), 3, 1.0),
    "j17_rf81_causal_s3": (dict(ARCHITECTURE="3,3,3,3", DISABLE_OPTIMIZATIONS=True, CAUSAL=True), 3, 1.0),
    # the dense-convolution ablation (rie.py:49-53: 2*pad+1 taps, stride 1), plain and causal
    "j17_rf27_dense_s3": (dict(ARCHITECTURE="3,3,3", DISABLE_OPTIMIZATIONS=True, DENSE=True), 3, 1.0),
    "j14_rf9_dense_causal_s2": (dict(ARCHITECTURE="3,3", NUM_KPTS=14, DISABLE_OPTIMIZATIONS=True, DENSE=True, CAUSAL=True, STAGE=2), 4, 1.0),
}


def load_synth(module, cfg, seed, out_scale):
    state = synth.synth_state(cfg, seed=seed, out_scale=out_scale)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in state.items()}
    module.load_state_dict(sd, strict=True)
    module.eval()
    return state


def run_with_taps(module, x, p, tap_names):
    taps, hooks = {}, []
    mods = dict(module.named_modules())
    for name in tap_names:
        def mk(nm):
            def hook(_m, _i, o):
                taps[nm] = o.detach().clone().numpy()
            return hook
        hooks.append(mods[name].register_forward_hook(mk(name)))
    with torch.no_grad():
        out = module(torch.from_numpy(x), torch.from_numpy(p)).numpy()
    for h in hooks:
        h.remove()
    return out, taps


def gen_models(only=None):
    for name, (over, batch, out_scale) in CASES.items():
        if only and name not in only:
            continue
        mc = default_model_config(**over)
        ref = RefModel(mc, {}, is_train=False)
        pos, trj = ref.get_pos_model(), ref.get_trj_model()
        cpos, ctrj = config_from_dicts(mc, "pos"), config_from_dicts(mc, "trj")
        load_synth(pos, cpos, 1, out_scale)
        load_synth(trj, ctrj, 2, out_scale)
        x = synth.synth_rays(batch, cpos, seed=3)
        p = synth.synth_param(batch, seed=4)
        nlev = len(cpos.filter_widths)
        pos_taps = (["LocalLayer_%s" % b for b in cpos.branch_names()] + ["GlobalInfo"]
                    + ["Integration_%s" % b for b in cpos.branch_names()])
        if cpos.stage != 1:
            pos_taps += ["FuseBlocks.%d" % i for i in range(5)]
        if cpos.camera_embedding:
            pos_taps += ["embedder"]
        # per-level pre-activation BN outputs, Torso only (clone happens before inplace LeakyReLU)
        # (the dilated form keeps every frame of every level: its per-level tensors are not the strided ones)
        per_level = not mc["DISABLE_OPTIMIZATIONS"]
        if per_level:
            pos_taps += ["LocalLayer_Torso.expand_bn"] + ["LocalLayer_Torso.layers_bn.%d" % i
                                                         for i in range(2 * (nlev - 1))]
        trj_taps = ["LocalLayer", "GlobalInfo", "Integration"]
        if per_level:
            trj_taps += ["LocalLayer.expand_bn"] + ["LocalLayer.layers_bn.%d" % i for i in range(2 * (nlev - 1))]
        if ctrj.camera_embedding:
            trj_taps += ["embedder"]
        out_pos, tp = run_with_taps(pos, x, p, pos_taps)
        out_trj, tt = run_with_taps(trj, x, p, trj_taps)
        blob = dict(x=x, param=p, out_pos=out_pos, out_trj=out_trj,
                    n_state_pos=np.int64(len(pos.state_dict())),
                    n_state_trj=np.int64(len(trj.state_dict())),
                    receptive_field=np.int64(pos.receptive_field()))
        for k, v in tp.items():
            # per-level taps: keep window 0 only to stay small; channels-first (C, T) as in torch
            blob["pos/" + k] = v[0] if ("_bn" in k) else v
        for k, v in tt.items():
            blob["trj/" + k] = v[0] if ("_bn" in k) else v
        blob["model_config_keys"] = np.array(sorted(mc.keys()))
        blob["model_config_vals"] = np.array([str(mc[k]) for k in sorted(mc.keys())])
        np.savez_compressed(os.path.join(HERE, "model_%s.npz" % name), **blob)
        print("model_%s: out_pos %s |max| %.3f  out_trj |max| %.3f  states %d/%d" % (
            name, out_pos.shape, np.abs(out_pos).max(), np.abs(out_trj).max(),
            len(pos.state_dict()), len(trj.state_dict())))


def ref_cameras():
    """(tag, K, R, t) exactly as the reference datasets hand them to CameraInfoPacket
    (lib/dataset/h36m_dataset.py:351-386, lib/dataset/mpii_3dhp_dataset.py:309-341)."""
    cams = []
    for subj, idxs in (("S9", (0, 1, 2, 3)), ("S1", (0,)), ("S11", (2,))):
        for i in idxs:
            ext, intr = h36m_cameras_extrinsic_params[subj][i], h36m_cameras_intrinsic_params[i]
            K = np.eye(3, dtype=np.float64)
            f32 = lambda v: np.array(v, dtype="float32")
            fl, ce = f32(intr["focal_length"]), f32(intr["center"])
            K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fl[0], fl[1], ce[0], ce[1]
            R = f32(ext["R"])
            t = np.array(f32(ext["translation"]) / 1000, dtype=np.float64).reshape(3, 1)
            cams.append(("h36m_%s_%d" % (subj, i), K, R, t))
    for subj in list(dhp_camera_params.keys())[:2]:
        for i, cam in enumerate(dhp_camera_params[subj][:3]):
            if "translation" not in cam:
                continue
            f32 = lambda v: np.array(v, dtype="float32")
            fl, ce = f32(cam["focal_length"]), f32(cam["center"])
            K = np.eye(3, dtype=np.float64)
            K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fl[0], fl[1], ce[0], ce[1]
            R = f32(cam["R"])
            t = np.array(f32(cam["translation"]), dtype=np.float64).reshape(3, 1)
            cams.append(("3dhp_%s_%d" % (subj, i), K, R, t))
    return cams


def gen_cameras():
    blob = {}
    tags = []
    for tag, K, R, t in ref_cameras():
        cam = CameraInfoPacket(P=None, K=K, R=R, t=t, res_w=1000, res_h=1000, azimuth=0,
                               dist_coeff=None, undistort=False)
        uv = 1000.0 * synth.hash_uniform("uv." + tag, (5, 17, 2), 7)
        rays = cam.get_cam_ray_given_uv(uv)
        X = np.concatenate([synth.hash_uniform("xw." + tag, (4, 17, 2), 7) * 2 - 1,
                            synth.hash_uniform("zw." + tag, (4, 17, 1), 7) * 1.8], axis=-1)
        tags.append(tag)
        blob.update({
            tag + "/K": K, tag + "/R": R.astype(np.float64), tag + "/t": t,
            tag + "/height": np.float64((-cam.Rw2c.T @ cam.Tw2c)[2][0]),
            tag + "/pitch": np.float64(cam.cam_pitch_rad),
            tag + "/Rc2n": cam.Rc2n, tag + "/Tc2n": cam.Tc2n,
            tag + "/Rn2w": cam.Rn2w, tag + "/Tn2w": cam.Tn2w,
            tag + "/Rw2n": cam.Rw2n, tag + "/Tw2n": cam.Tw2n,
            tag + "/uv": uv, tag + "/rays": rays,
            tag + "/uv_back": cam.get_uv_given_cam_ray(rays),
            tag + "/Xw": X, tag + "/Xn": cam.world2normalized(X),
            tag + "/Xw_back": cam.normalized2world(cam.world2normalized(X)),
            tag + "/proj": cam.project(np.concatenate([X, np.ones_like(X[..., :1])], -1)),
        })
    blob["tags"] = np.array(tags)
    np.savez_compressed(os.path.join(HERE, "cameras.npz"), **blob)
    print("cameras:", len(tags))


def synth_clip(tag, n, cam, seed):
    """World joints ~ N(0,0.3^2)+[0,0,1] with temporal smoothness, projected by `cam`."""
    u = synth.hash_uniform("clip." + tag, (n, 17, 3, 4), seed)
    g = (u.sum(-1) - 2.0) * np.sqrt(3.0)            # ~N(0,1) (Irwin-Hall, 4 terms)
    pose = 0.3 * g[:1] + 0.03 * np.cumsum(g, axis=0) / np.sqrt(np.arange(1, n + 1))[:, None, None]
    Xw = pose + np.array([0.0, 0.0, 1.0])
    uv = cam.project(np.concatenate([Xw, np.ones_like(Xw[..., :1])], -1))
    rays = cam.get_cam_ray_given_uv(uv)
    gt = cam.world2normalized(Xw)
    return Xw, uv, rays, gt


def gen_evalcore():
    mc = default_model_config(ARCHITECTURE="3,3,3")
    ref = RefModel(mc, {}, is_train=False)
    pos, trj = ref.get_pos_model(), ref.get_trj_model()
    cpos, ctrj = config_from_dicts(mc, "pos"), config_from_dicts(mc, "trj")
    load_synth(pos, cpos, 1, 1.0)
    load_synth(trj, ctrj, 2, 1.0)
    cams_all = ref_cameras()
    picks = [(cams_all[0], 100), (cams_all[1], 57), (cams_all[6], 31)]
    cams, p3d, p2d, blob = [], [], [], {}
    for ci, ((tag, K, R, t), n) in enumerate(picks):
        cam = CameraInfoPacket(P=None, K=K, R=R, t=t, res_w=1000, res_h=1000, azimuth=0,
                               dist_coeff=None, undistort=False)
        Xw, uv, rays, gt = synth_clip(tag, n, cam, 11 + ci)
        cams.append(cam)
        p3d.append(gt.astype(np.float32))          # dataset stores normalized-space GT as float32
        p2d.append(rays.astype(np.float32))
        blob.update({"clip%d/K" % ci: K, "clip%d/R" % ci: R.astype(np.float64), "clip%d/t" % ci: t,
                     "clip%d/uv" % ci: uv, "clip%d/rays" % ci: rays.astype(np.float32),
                     "clip%d/gt_norm" % ci: gt.astype(np.float32), "clip%d/Xw" % ci: Xw})
    # H36M 17-joint symmetry (lib/dataset/h36m_dataset.py skeleton after joint removal)
    kps_left, kps_right = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
    data_config = {"RAY_ENCODING": True}
    tr = Trainer(data_config, mc, {"LEARNING_RATE": 1e-3}, {}, None, None,
                 {"train_pos": pos, "test_pos": pos, "train_trj": trj, "test_trj": trj},
                 None, kps_left, kps_right, kps_left, kps_right, None)
    pad = (cpos.receptive_field - 1) // 2
    for flip in (False, True):
        gen = UnchunkedGenerator(cams, p3d, p2d, pad=pad, causal_shift=0, kps_left=kps_left,
                                 kps_right=kps_right, joints_left=kps_left, joints_right=kps_right)
        e = tr.evaluate_core(gen, flip_test=flip)
        blob["metrics_flip%d" % int(flip)] = np.array(e, dtype=np.float64)
        print("evaluate_core flip=%s ->" % flip, e)
        for ci in range(len(cams)):
            g1 = UnchunkedGenerator([cams[ci]], [p3d[ci]], [p2d[ci]], pad=pad, causal_shift=0,
                                    kps_left=kps_left, kps_right=kps_right,
                                    joints_left=kps_left, joints_right=kps_right)
            blob["clip%d/metrics_flip%d" % (ci, int(flip))] = np.array(
                tr.evaluate_core(g1, flip_test=flip), dtype=np.float64)
    gen = UnchunkedGenerator([cams[0]], [p3d[0]], [p2d[0]], pad=pad, causal_shift=0)
    blob["clip0/predictions"] = tr.evaluate_core(gen, return_predictions=True)
    blob["kps_left"], blob["kps_right"] = np.array(kps_left), np.array(kps_right)
    np.savez_compressed(os.path.join(HERE, "evalcore.npz"), **blob)


def gen_dataset():
    """The reference's Data front end (lib/dataset/__init__.py) on a small synthetic archive pair in the 3DHP
    format (plain per-camera keypoint arrays; the TS1 test camera of lib/dataset/mpii_3dhp_dataset.py), and the
    joints its H36M class keeps of the 32-joint mocap skeleton."""
    import copy
    import tempfile
    from lib.dataset import Data
    from lib.dataset.h36m_dataset import h36m_skeleton
    rng = np.random.default_rng(11)
    cam_tab = dhp_camera_params["TS1"][0]
    K = np.eye(3)
    K[0, 0], K[1, 1] = np.array(cam_tab["focal_length"], dtype="float32")
    K[0, 2], K[1, 2] = np.array(cam_tab["center"], dtype="float32")
    R = np.array(cam_tab["R"], dtype="float32").astype(np.float64)
    t = np.array(cam_tab["translation"], dtype="float32").astype(np.float64).reshape(3, 1)
    lengths = {"Seq A 1": 31, "Seq A 2": 17, "Other": 23}
    pos3d, pos2d = {"TS1": {}}, {"TS1": {}}
    for act, n in lengths.items():
        # world points in front of the camera: camera-frame box pushed back through the extrinsics
        pc = np.stack([rng.uniform(-0.8, 0.8, (n, 17)), rng.uniform(-0.9, 0.9, (n, 17)), rng.uniform(3.0, 5.0, (n, 17))], -1)
        pw = (pc - t.T) @ R                                     # Xw = R^T (Xc - t)
        pos3d["TS1"][act] = pw.astype(np.float32)
        uv = np.stack([pc[..., 0] / pc[..., 2] * K[0, 0] + K[0, 2], pc[..., 1] / pc[..., 2] * K[1, 1] + K[1, 2]], -1)
        uv = np.concatenate([uv, uv[-1:].repeat(4, 0)], 0) + rng.normal(0, 1.0, (n + 4, 17, 2))   # 4 extra video frames
        pos2d["TS1"][act] = [uv.astype(np.float32)]
    meta = {"layout_name": "3dhp", "num_joints": 17, "keypoints_symmetry": [[4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]]}
    with tempfile.TemporaryDirectory() as d:
        p3, p2 = os.path.join(d, "d3.npz"), os.path.join(d, "d2.npz")
        np.savez_compressed(p3, positions_3d=pos3d)
        np.savez_compressed(p2, positions_2d=pos2d, metadata=meta)
        data = Data({"DATASET": "3dhp", "WORLD_3D_GT_EVAL": True, "KEYPOINTS": "gt", "REMOVE_IRRELEVANT_KPTS": False,
                     "GT_3D": p3, "GT_2D": p2, "FRAME_PATH": d, "INTRINSIC_ENCODING": False, "RAY_ENCODING": True,
                     "DOWNSAMPLE": 1})
        blob = dict(table_R=np.array(cam_tab["R"], dtype=np.float64), table_translation=np.array(cam_tab["translation"], dtype=np.float64),
                    table_focal_length=np.array(cam_tab["focal_length"], dtype=np.float64),
                    table_center=np.array(cam_tab["center"], dtype=np.float64),
                    actions=np.array(list(lengths.keys())))
        kl, kr = data.get_2d_kpts()
        jl, jr = data.get_3d_joints()
        blob.update(kps_left=np.array(kl), kps_right=np.array(kr), joints_left=np.array(jl), joints_right=np.array(jr))
        for i, act in enumerate(lengths):
            blob["in3d/%d" % i], blob["in2d/%d" % i] = pos3d["TS1"][act], pos2d["TS1"][act][0]
            cams, p3d, p2d = data.fetch_via_action([("TS1", act)])
            assert len(p3d) == 1
            blob["gt_norm/%d" % i], blob["rays/%d" % i] = p3d[0], p2d[0]
            c0 = cams[0][0] if isinstance(cams[0], (list, tuple)) else cams[0]
            blob["pitch/%d" % i] = np.float64(c0.cam_pitch_rad)
    sk = copy.deepcopy(h36m_skeleton)
    blob["h36m_kept_17"] = np.array(sk.remove_joints([4, 5, 9, 10, 11, 16, 20, 21, 22, 23, 24, 28, 29, 30, 31]))
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **blob)
    print("dataset: clips", {k: v.shape for k, v in blob.items() if k.startswith("rays/")}, "kept", blob["h36m_kept_17"])


def gen_frontends():
    """The other two camera front ends of the reference, on small synthetic archives:
    (a) the camera-augmented H36M set (lib/dataset/h36m_aug_dataset.py through lib/dataset/__init__.py ``Data``): a
        JSON camera list shared by the original and the scaled subjects, 32-joint mocap archive reduced to 17 joints,
        per-camera fetches as ``CAMERA_WISE_PERFORMANCE`` does them;
    (b) HumanEva-I (lib/dataset/humaneva_dataset.py): the camera tables registered under 'Train/' and 'Validate/'
        prefixes, the 15-joint symmetry lists and the universal 14-joint selection.  Its cameras are built with
        undistort=True, whose only effect in the constructor is ``pp_cam`` (through cv2, absent here: the stub below
        returns the points unchanged) - no number stored here depends on it."""
    import json
    import tempfile
    from lib.dataset import Data
    rng = np.random.default_rng(23)
    blob = {}
    # ---------------- (a) h36m_aug
    cams_json = []
    for i in range(3):
        e = dict(h36m_cameras_intrinsic_params[i])
        ext = h36m_cameras_extrinsic_params["S9"][i]
        e["R"] = np.array(ext["R"], dtype=np.float64).tolist()
        e["translation"] = (np.array(ext["translation"], dtype=np.float64).reshape(3, 1) / 1000.0).tolist()
        cams_json.append(e)
    subjects = ["S9", "S9_0.9"]
    acts = {"Walk 1": 14, "Sit": 9}
    pos3d, pos2d = {}, {}
    keep17 = [0, 1, 2, 3, 6, 7, 8, 12, 13, 14, 15, 17, 18, 19, 25, 26, 27]
    for s_i, sub in enumerate(subjects):
        pos3d[sub], pos2d[sub] = {}, {}
        for act, n in acts.items():
            pw = rng.normal(0, 0.35, (n, 32, 3)) * (0.9 if s_i else 1.0) + np.array([0.2, -0.1, 0.95])
            pos3d[sub][act] = pw.astype(np.float32)
            views = []
            for e in cams_json:
                R, t = np.array(e["R"]), np.array(e["translation"]).reshape(3, 1)
                pc = pw[:, keep17] @ R.T + t.T
                uv = np.stack([pc[..., 0] / pc[..., 2] * e["focal_length"][0] + e["center"][0],
                               pc[..., 1] / pc[..., 2] * e["focal_length"][1] + e["center"][1]], -1)
                views.append((uv + rng.normal(0, 0.5, uv.shape)).astype(np.float32))
            pos2d[sub][act] = views
    meta = {"layout_name": "h36m", "num_joints": 17, "keypoints_symmetry": [[4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]]}
    with tempfile.TemporaryDirectory() as d:
        p3, p2, pj = os.path.join(d, "d3.npz"), os.path.join(d, "d2.npz"), os.path.join(d, "cams.json")
        np.savez_compressed(p3, positions_3d=pos3d)
        np.savez_compressed(p2, positions_2d=pos2d, metadata=meta)
        json.dump(cams_json, open(pj, "w"), indent=4)
        data = Data({"DATASET": "h36m_aug", "CAMERA_PARAM": pj, "CAMERA_WISE_PERFORMANCE": True, "WORLD_3D_GT_EVAL": True,
                     "KEYPOINTS": "gt", "REMOVE_IRRELEVANT_KPTS": False, "GT_3D": p3, "GT_2D": p2, "FRAME_PATH": d,
                     "INTRINSIC_ENCODING": False, "RAY_ENCODING": True, "DOWNSAMPLE": 1})
        ds = data.get_dataset()
        blob["aug/cams_json"] = np.array(json.dumps(cams_json))
        blob["aug/camera_dist"] = np.array([str(c) for c in ds.camera_dist])
        blob["aug/subjects_all"] = np.array(list(ds.camera_info.keys()))
        blob["aug/subjects"] = np.array(subjects)
        blob["aug/actions"] = np.array(list(acts.keys()))
        for s_i, sub in enumerate(subjects):
            for a_i, act in enumerate(acts):
                blob["aug/in3d/%d/%d" % (s_i, a_i)] = pos3d[sub][act]
                for ci in range(3):
                    blob["aug/in2d/%d/%d/%d" % (s_i, a_i, ci)] = pos2d[sub][act][ci]
                    cams, p3d, p2d = data.fetch_via_action([(sub, act)], camera_idx=ci)
                    assert len(p3d) == 1 and len(p2d) == 1
                    blob["aug/gt_norm/%d/%d/%d" % (s_i, a_i, ci)] = p3d[0]
                    blob["aug/rays/%d/%d/%d" % (s_i, a_i, ci)] = p2d[0]
        for ci in range(3):
            c = ds.camera_info["S9_0.9"][ci]
            blob["aug/cam%d/Rn2w" % ci], blob["aug/cam%d/Tn2w" % ci] = c.Rn2w, c.Tn2w
            blob["aug/cam%d/height_pitch" % ci] = np.array([c.cam_orig_world.reshape(-1)[2], c.cam_pitch_rad])
    # ---------------- (b) HumanEva
    import cv2 as cv2_stub
    cv2_stub.undistortPoints = lambda pts, K, dist, P=None: pts          # only feeds pp_cam, which nothing here reads
    from lib.dataset.humaneva_dataset import (HumanEvaDataset, humaneva_cameras_extrinsic_params,
                                              humaneva_cameras_intrinsic_params, humaneva_skeleton)
    with tempfile.TemporaryDirectory() as d:
        p3 = os.path.join(d, "he3.npz")
        np.savez_compressed(p3, positions_3d={"Train/S1": {"Walking 1 chunk0": rng.normal(0, 0.3, (5, 15, 3)).astype(np.float32)}})
        he = HumanEvaDataset(p3)
        keys = list(he.camera_info.keys())
        blob["he/keys"] = np.array(keys)
        blob["he/ext_json"] = np.array(json.dumps(humaneva_cameras_extrinsic_params))
        blob["he/int_json"] = np.array(json.dumps(humaneva_cameras_intrinsic_params))
        for k in keys:
            c = he.camera_info[k][0]
            tag = "he/" + k.replace("/", "_")
            blob[tag + "/K"], blob[tag + "/Rn2w"], blob[tag + "/Tn2w"] = c.K, c.Rn2w, c.Tn2w
            blob[tag + "/height_pitch"] = np.array([c.cam_orig_world.reshape(-1)[2], c.cam_pitch_rad])
            blob[tag + "/dist"] = c.dist_coeff
            blob[tag + "/n"] = np.array(len(he.camera_info[k]))
        blob["he/joints_left"] = np.array(humaneva_skeleton.joints_left())
        blob["he/joints_right"] = np.array(humaneva_skeleton.joints_right())
        kp = {"positions_2d": np.array({"Train/S1": {"A": [np.arange(2 * 15 * 2, dtype=np.float32).reshape(2, 15, 2)]}}, dtype=object),
              "metadata": np.array({"layout_name": "humaneva15", "num_joints": 15, "keypoints_symmetry": [[2, 3, 4, 8, 9, 10], [5, 6, 7, 11, 12, 13]]}, dtype=object)}
        upd, upd_meta = HumanEvaDataset.remove_irrelevant_kpts(kp, universal=True)
        blob["he/universal_in"] = kp["positions_2d"].item()["Train/S1"]["A"][0]
        blob["he/universal_out"] = upd["Train/S1"]["A"][0]
        blob["he/universal_symmetry"] = np.array(upd_meta["keypoints_symmetry"])
    # ---------------- (c) the synthetic camera sets: data/camera_augmentation.py's own functions on the base camera
    # the script uses (S1, camera 1), for a spread of the 'Train' grid (yaw x distance ratio x pitch)
    sys.modules.setdefault("ipdb", types.ModuleType("ipdb"))
    try:
        import matplotlib
        matplotlib.use("Agg")
    except Exception:
        sys.modules.setdefault("matplotlib", types.ModuleType("matplotlib"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_camera_augmentation", os.path.join(REF, "data", "camera_augmentation.py"))
    ca = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ca)
    ext = h36m_cameras_extrinsic_params["S1"][1]
    R0 = np.array(ext["R"], dtype="float32").astype(np.float64)
    T0 = (np.array(ext["translation"], dtype="float32") / 1000).astype(np.float64).reshape(3, 1)
    center = np.array([0, 0, 1.8]).reshape(3, 1)
    combos = [(60, 2.0, -26), (180, 2.4, -10), (300, 3.0, 10), (60, 2.8, 0), (180, 2.0, 4), (300, 2.2, -18)]
    blob["grid/R0"], blob["grid/T0"], blob["grid/combos"] = R0, T0, np.array(combos, dtype=np.float64)
    for i, (yaw_d, ratio, pitch_d) in enumerate(combos):
        T1 = ca.camera_translation(T0, center, ratio)
        R2, T2 = ca.rotate_camera(R0, T1, center, np.array(ca.AXIS_Z), ca.convertdegree2euler(yaw_d))
        pos = -np.dot(R2.T, T2)
        axis = np.array([-pos[1][0], pos[0][0], 0])
        R3, T3 = ca.rotate_camera(R2, T2, center, axis, ca.convertdegree2euler(pitch_d))
        blob["grid/R/%d" % i], blob["grid/T/%d" % i] = R3, T3
    np.savez_compressed(os.path.join(HERE, "frontends.npz"), **blob)
    print("frontends: aug subjects", len(blob["aug/subjects_all"]), "camera_dist", list(blob["aug/camera_dist"]), "| humaneva keys", keys)


def gen_cameras_3dhp():
    """The 14 cameras of MPI-INF-3DHP's S1/Seq1 (lib/dataset/mpii_3dhp_dataset.py:9-251; BASELINE configs[3] draws one
    per window): CameraInfoPacket constants and uv -> ray pairs, built exactly as Mpii3dhpDataset.__init__ builds them
    (:309-341: float32 table values, undistort=False).  A file of its own: cameras.npz stays byte-identical."""
    blob, tags = {}, []
    f32 = lambda v: np.array(v, dtype="float32")
    for i in range(14):
        cam = dhp_camera_params["S1_Seq1_%d" % i][0]
        fl, ce = f32(cam["focal_length"]), f32(cam["center"])
        K = np.eye(3, dtype=np.float64)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fl[0], fl[1], ce[0], ce[1]
        R = f32(cam["R"])
        t = np.array(f32(cam["translation"]), dtype=np.float64).reshape(3, 1)
        pkt = CameraInfoPacket(P=None, K=K, R=R, t=t, res_w=cam["res_w"], res_h=cam["res_h"], azimuth=cam["azimuth"],
                               dist_coeff=None, undistort=False)
        tag = "3dhp_S1_Seq1_%d" % i
        uv = 2048.0 * synth.hash_uniform("uv." + tag, (3, 17, 2), 9)
        tags.append(tag)
        blob.update({tag + "/K": K, tag + "/R": R.astype(np.float64), tag + "/t": t,
                     tag + "/height": np.float64((-pkt.Rw2c.T @ pkt.Tw2c)[2][0]),
                     tag + "/pitch": np.float64(pkt.cam_pitch_rad),
                     tag + "/Rn2w": pkt.Rn2w, tag + "/Tn2w": pkt.Tn2w,
                     tag + "/uv": uv, tag + "/rays": pkt.get_cam_ray_given_uv(uv)})
    blob["tags"] = np.array(tags)
    np.savez_compressed(os.path.join(HERE, "cameras_3dhp.npz"), **blob)
    print("cameras_3dhp:", len(tags))


def gen_undistort():
    """What CAN be pinned of the undistortion (lib/camera/camera.py:412-421 calls cv2.undistortPoints, which is absent):
    the reference's own FORWARD model, distortPoint (data/camera_augmentation.py:502-542), on a pixel grid for the four
    H36M coefficient sets as lib/dataset/h36m_dataset.py:378-380 orders them.  undistort(distortPoint(p)) must return p."""
    for name in ("ipdb", "h5py", "mat73", "cdflib"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, os.path.join(REF, "data"))
    import matplotlib
    matplotlib.use("Agg")
    from data.camera_augmentation import distortPoint
    blob = {}
    for i, intr in enumerate(h36m_cameras_intrinsic_params):
        f32 = lambda v: np.array(v, dtype="float32")
        fl, ce = f32(intr["focal_length"]), f32(intr["center"])
        K = np.eye(3, dtype=np.float64)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fl[0], fl[1], ce[0], ce[1]
        rad, tan = f32(intr["radial_distortion"]), f32(intr["tangential_distortion"])
        dist = np.concatenate((rad[:2], tan, rad[2:])).astype(np.float32).reshape(5)      # h36m_dataset.py:378-380
        gx, gy = np.meshgrid(np.linspace(20.0, float(intr["res_w"]) - 20.0, 13), np.linspace(20.0, float(intr["res_h"]) - 20.0, 13))
        ideal = np.stack([gx.ravel(), gy.ravel()], axis=0)                                 # (2, N) as distortPoint expects
        distorted = distortPoint(ideal.copy(), K, dist.reshape(1, 5).astype(np.float64))
        blob.update({"cam%d/K" % i: K, "cam%d/dist" % i: dist.astype(np.float64), "cam%d/ideal" % i: ideal.T.copy(),
                     "cam%d/distorted" % i: distorted.T.copy()})
        # -- the ALGORITHM pin (OpenCV itself cannot be run here): the reference's forward model on a DENSE grid, and what
        # cv2.undistortPoints(pts, K, dist, P=K) of opencv-python 4.4.0.42 (requirements.txt:40; call site
        # lib/camera/camera.py:420) is documented to compute for it - cvUndistortPointsInternal with the default
        # TermCriteria(MAX_ITER, 5, 0.01): exactly five iterations of x <- (x0 - tangential(x)) / radial(x) on normalised
        # coordinates, re-projected with P = K - restated here independently of the product and of oracle/, in float64.
        # `dense_und5` is that result point by point, `res5_*` the residual those five iterations leave against the true
        # inverse (the grid the reference's distortPoint started from), `res200_max` the same after 200 iterations: the
        # iteration converges to the grid, five iterations stop where the table says.
        dx_, dy_ = np.meshgrid(np.linspace(0.0, float(intr["res_w"]), 65), np.linspace(0.0, float(intr["res_h"]), 65))
        dense = np.stack([dx_.ravel(), dy_.ravel()], axis=0)
        dense_d = distortPoint(dense.copy(), K, dist.reshape(1, 5).astype(np.float64))

        def cv_undistort(pts, iters):
            k1, k2, p1, p2, k3 = (float(v) for v in dist.astype(np.float64))
            x0 = (pts[0] - K[0, 2]) / K[0, 0]
            y0 = (pts[1] - K[1, 2]) / K[1, 1]
            x, y = x0.copy(), y0.copy()
            for _ in range(iters):
                r2 = x * x + y * y
                icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
                deltaX = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
                deltaY = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
                x = (x0 - deltaX) * icdist
                y = (y0 - deltaY) * icdist
            return np.stack([x * K[0, 0] + K[0, 2], y * K[1, 1] + K[1, 2]], axis=0)
        und5, und200 = cv_undistort(dense_d, 5), cv_undistort(dense_d, 200)
        w_, h_ = float(intr["res_w"]), float(intr["res_h"])
        inner = (np.abs(dense[0] - 0.5 * w_) <= 0.4 * w_) & (np.abs(dense[1] - 0.5 * h_) <= 0.4 * h_)
        r5 = np.abs(und5 - dense).max(axis=0)
        blob.update({"cam%d/dense_ideal" % i: dense.T.copy(), "cam%d/dense_distorted" % i: dense_d.T.copy(),
                     "cam%d/dense_und5" % i: und5.T.copy(), "cam%d/dense_inner" % i: inner,
                     "cam%d/res5_inner_max" % i: np.float64(r5[inner].max()), "cam%d/res5_all_max" % i: np.float64(r5.max()),
                     "cam%d/res200_max" % i: np.float64(np.abs(und200 - dense).max())})
    blob["n"] = np.int64(len(h36m_cameras_intrinsic_params))
    np.savez_compressed(os.path.join(HERE, "undistort.npz"), **blob)
    print("undistort: %d coefficient sets, %d points each" % (int(blob["n"]), ideal.shape[1]))


def gen_losses():
    a = (synth.hash_uniform("loss.a", (6, 1, 17, 3), 5) * 2 - 1)
    b = a + 0.1 * (synth.hash_uniform("loss.b", (6, 1, 17, 3), 5) * 2 - 1)
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    blob = dict(pred=a, target=b,
                mpjpe=np.float64(ref_loss.mpjpe(ta, tb).item()),
                n_mpjpe=np.float64(ref_loss.n_mpjpe(ta, tb).item()),
                p_mpjpe=np.float64(ref_loss.p_mpjpe(a.reshape(-1, 17, 3).copy(), b.reshape(-1, 17, 3).copy())),
                mpjve=np.float64(ref_loss.mean_velocity_error(a.reshape(-1, 17, 3), b.reshape(-1, 17, 3))))
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **blob)
    print("losses:", {k: float(v) for k, v in blob.items() if np.ndim(v) == 0})


if __name__ == "__main__":
    if len(sys.argv) > 1:                      # python make_golden.py <case> ... | dataset: only those
        if "dataset" in sys.argv[1:]:
            gen_dataset()
        if "frontends" in sys.argv[1:]:
            gen_frontends()
        if "cameras_3dhp" in sys.argv[1:]:
            gen_cameras_3dhp()
        if "undistort" in sys.argv[1:]:
            gen_undistort()
        gen_models(only=set(sys.argv[1:]) - {"dataset", "frontends", "cameras_3dhp", "undistort"} or {"-"})
        sys.exit(0)
    gen_cameras()
    gen_cameras_3dhp()
    gen_undistort()
    gen_losses()
    gen_models()
    gen_evalcore()
    gen_dataset()
    gen_frontends()
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print("total fixture bytes: %.2f MB" % (tot / 1e6))
