"""Batched evaluation over clips: the caller side of the lifting path.

Reproduces the behaviour of ``Trainer.evaluate_core`` / ``Trainer.evaluate``
(lib/train_val/trainer.py:283-405, :407-483) around the HIP forward:

* a clip is edge-padded by (RF-1)//2 frames per side (lib/dataloader/generators.py:213-216) and
  window i covers padded frames [i, i+RF) (trainer.py:47-58) - the windows are gathered inside
  the first-layer tiles (`Ray3DLifter.forward_clip`), not materialised;
* the camera row [height, pitch] is shared by all windows of the clip (trainer.py:297,324);
* prediction = pos + trj, optionally averaged with the mirrored pass (trainer.py:299-302,338-353);
* prediction and ground truth go to world coordinates in float64 (trainer.py:355-364) and the five
  per-clip errors are accumulated weighted by the number of frames (trainer.py:386-403).

Multi-GPU: whole clips are partitioned over ranks (one process per GPU), each rank evaluates its
share with resident weights, and ONE all_gather of the fixed-size per-clip partial rows
(RCCL over xGMI with the nccl backend, gloo on CPU) brings them together.  There is no other
collective on the path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import metrics as M
from .camera import Camera

PARTIAL_COLS = 8   # clip_id, action_id, n_frames, sum_mpjpe, sum_pmpjpe, sum_nmpjpe, sum_vel, sum_root


@dataclass
class Clip:
    camera: Camera
    rays: np.ndarray          # (N, J, F) float32 ray-encoded keypoints (model input space)
    gt_norm: np.ndarray       # (N, J, 3) float32 ground truth in the normalised frame
    action: str = ""
    clip_id: int = 0


def pad_clip(rays: np.ndarray, pad: int, causal_shift: int = 0) -> np.ndarray:
    """np.pad(seq, ((pad + shift, pad - shift), (0,0), (0,0)), 'edge') - generators.py:213-216;
    shift = pad for CAUSAL models (main.py:85-89: the window then ends at the frame it predicts)."""
    return np.concatenate([np.repeat(rays[:1], pad + causal_shift, axis=0), rays,
                           np.repeat(rays[-1:], pad - causal_shift, axis=0)], axis=0)


def mirror_input(clip: torch.Tensor, kps_left: Sequence[int], kps_right: Sequence[int]) -> torch.Tensor:
    """trainer.py:299-302: negate x, swap left/right keypoints."""
    out = clip.clone()
    out[..., 0] *= -1
    out[:, list(kps_left) + list(kps_right)] = out[:, list(kps_right) + list(kps_left)]
    return out


def mirror_output(pred: torch.Tensor, joints_left: Sequence[int], joints_right: Sequence[int]) -> torch.Tensor:
    """trainer.py:340-342 on (N,1,J,3) predictions."""
    out = pred.clone()
    out[..., 0] *= -1
    out[:, :, list(joints_left) + list(joints_right)] = out[:, :, list(joints_right) + list(joints_left)]
    return out


def predict_clip(lift_clip: Callable, clip: Clip, rf: int, device, flip: bool = False,
                 kps_left: Sequence[int] = (), kps_right: Sequence[int] = (), causal: bool = False,
                 joints_left: Optional[Sequence[int]] = None, joints_right: Optional[Sequence[int]] = None) -> torch.Tensor:
    """(N,1,J,3) absolute poses in the normalised frame for one clip.
    `lift_clip(padded (N+RF-1,J,F) tensor, param_row (E,) tensor) -> (N,1,J,3)`."""
    pad = (rf - 1) // 2
    padded = torch.from_numpy(pad_clip(np.asarray(clip.rays, dtype=np.float32), pad, pad if causal else 0)).to(device)
    prow = torch.from_numpy(clip.camera.param()).to(device)
    pred = lift_clip(padded, prow)
    if flip:
        pred_m = lift_clip(mirror_input(padded, kps_left, kps_right), prow)
        pred = 0.5 * (pred + mirror_output(pred_m, kps_left if joints_left is None else joints_left,
                                           kps_right if joints_right is None else joints_right))
    return pred


def partial_rows(headers: Sequence[tuple], device) -> torch.Tensor:
    """The (k, PARTIAL_COLS) float64 matrix of a rank's per-clip rows with the host-known columns - clip id, action id,
    frame count - filled in: ONE host-to-device copy per evaluation instead of one per clip (the error columns are written
    on the device by :func:`clip_partials_hip`)."""
    rows = torch.zeros((len(headers), PARTIAL_COLS), dtype=torch.float64)
    if headers:
        rows[:, :3] = torch.tensor([[float(v) for v in h] for h in headers], dtype=torch.float64)
    return rows.to(device)


def clip_partials_hip(pred_norm: torch.Tensor, clip: Clip, action_id: int = 0,
                      gt_dev: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """clip_partials on the GPU through r3d_clip_metrics: world transform, the five error sums and the per-frame
    Procrustes fits in one float64 kernel on the current stream - no D2H copy of the predictions.
    `gt_dev`: the clip's ground truth already on the device (callers that keep a data set resident in HBM).
    `out`: a row of :func:`partial_rows` (header columns already on the device): only the five sums are written, by a
    device copy - no host tensor, no host synchronisation per clip, whatever the number of ranks."""
    from . import _capi
    dev = pred_norm.device
    n = pred_norm.shape[0]
    pred = pred_norm.reshape(n, -1, 3).contiguous().float()
    if gt_dev is not None:
        gt = gt_dev.to(dev, torch.float32).reshape(n, -1, 3).contiguous()
    else:
        gt = torch.from_numpy(np.ascontiguousarray(clip.gt_norm, dtype=np.float32)).to(dev, non_blocking=True).reshape(n, -1, 3)
    assert gt.shape == pred.shape, "ground truth %s vs prediction %s" % (tuple(gt.shape), tuple(pred.shape))
    sums = torch.empty(_capi.METRIC_OUT_DOUBLES, dtype=torch.float64, device=dev)
    _capi.clip_metrics(pred.data_ptr(), gt.data_ptr(), n, pred.shape[1], np.asarray(clip.camera.Rn2w, dtype=np.float64),
                       np.asarray(clip.camera.Tn2w, dtype=np.float64).reshape(3), sums.data_ptr(),
                       torch.cuda.current_stream(dev).cuda_stream)
    if out is not None:
        assert out.shape == (PARTIAL_COLS,) and out.dtype == torch.float64 and out.device == dev
        out[3:8] = sums[:5]           # R3D_METRIC_* order == columns 3..7 (mpjpe, p-mpjpe, n-mpjpe, velocity, root)
        return out
    row = torch.empty(PARTIAL_COLS, dtype=torch.float64, device=dev)
    for c, v in enumerate((clip.clip_id, action_id, n)):
        row[c].fill_(float(v))        # (scalars travel as kernel arguments: no pageable host tensor, no blocking copy)
    row[3:8] = sums[:5]
    return row


def clip_partials(pred_norm: torch.Tensor, clip: Clip, action_id: int = 0) -> torch.Tensor:
    """One PARTIAL_COLS row (float64, on pred's device): N-weighted error sums in metres.  Predictions on a GPU
    go through the HIP kernel; CPU tensors (host-logic tests with a stand-in lifter) through torch."""
    if pred_norm.is_cuda:
        return clip_partials_hip(pred_norm, clip, action_id)
    dev = pred_norm.device
    n = pred_norm.shape[0]
    R = torch.from_numpy(clip.camera.Rn2w.T.copy()).to(dev)
    T = torch.from_numpy(clip.camera.Tn2w.T.copy()).to(dev)
    pw = pred_norm.to(torch.float64).reshape(n, 1, -1, 3) @ R + T             # trainer.py:358
    gw = torch.from_numpy(np.asarray(clip.gt_norm, dtype=np.float32)).to(dev).to(torch.float64)
    gw = gw.reshape(n, 1, -1, 3) @ R + T                                        # trainer.py:359
    row = torch.zeros(PARTIAL_COLS, dtype=torch.float64, device=dev)
    row[0], row[1], row[2] = clip.clip_id, action_id, n
    row[3] = n * M.mpjpe(pw, gw)                                                # trainer.py:386
    row[4] = n * M.p_mpjpe(pw.reshape(n, -1, 3), gw.reshape(n, -1, 3))          # :393
    row[5] = n * M.n_mpjpe(pw, gw)                                              # :388
    row[6] = n * M.mean_velocity_error(pw.reshape(n, -1, 3), gw.reshape(n, -1, 3))   # :395
    row[7] = n * M.mpjpe(pw[:, :, 0:1], gw[:, :, 0:1])                          # :387
    return row


def reduce_partials(rows: torch.Tensor) -> Dict[int, tuple]:
    """{action_id: (e1, e2, e3, ev, er)} in millimetres - trainer.py:399-403 per action."""
    rows = rows.detach().to("cpu", torch.float64)
    out = {}
    for a in sorted(set(int(v) for v in rows[:, 1].tolist())):
        sel = rows[rows[:, 1] == a]
        n = sel[:, 2].sum()
        out[a] = tuple(float(sel[:, c].sum() / n * 1000.0) for c in (3, 4, 5, 6, 7))
    return out


def reduce_camera_wise(rows: torch.Tensor, camera_index: Sequence[int], camera_ids: Sequence[str]) -> List[tuple]:
    """``CAMERA_WISE_PERFORMANCE`` (lib/train_val/trainer.py:425-446) from the gathered per-clip rows: for every camera
    of the list (file order) and every action, the frame-weighted errors of THAT camera's clips (what
    ``fetch_via_action(..., camera_idx=cam_idx)`` + ``evaluate_core`` produce); after each camera the reference logs
    the mean over all (camera, action) errors collected SO FAR - its lists are not reset between cameras - rounded to
    0.1 mm.  ``camera_index[clip_id]`` is the clip's camera position (``PoseData.camera_index``).
    Returns ``[(camera id, (p1, p2, p3, vel, root))]`` in that cumulative form; :func:`format_camera_report` prints it."""
    rows = rows.detach().to("cpu", torch.float64)
    cam_of = torch.tensor([int(camera_index[int(c)]) for c in rows[:, 0].tolist()], dtype=torch.int64)
    collected: List[List[float]] = []
    out = []
    for ci, cid in enumerate(camera_ids):
        sel_c = rows[cam_of == ci]
        for a in sorted(set(int(v) for v in sel_c[:, 1].tolist())):
            sel = sel_c[sel_c[:, 1] == a]
            n = sel[:, 2].sum()
            collected.append([float(sel[:, c].sum() / n * 1000.0) for c in (3, 4, 5, 6, 7)])
        if collected:
            out.append((str(cid), tuple(round(float(v), 1) for v in np.mean(np.array(collected), axis=0))))
    return out


def format_camera_report(per_camera: Sequence[tuple]) -> List[str]:
    """'CAM ID <id>, p1 p2 p3 vel root' lines (trainer.py:446)."""
    return ["CAM ID %s, %s %s %s %s %s" % ((cid,) + tuple(v)) for cid, v in per_camera]


def action_average(per_action: Dict[int, tuple]) -> tuple:
    """Unweighted mean over actions, rounded to 0.1 mm like trainer.py:473-477."""
    arr = np.array(list(per_action.values()), dtype=np.float64)
    return tuple(round(float(v), 1) for v in arr.mean(axis=0))


# ------------------------------------------------------------------------------------ sharding

def shard_clips(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy longest-first bin packing of whole clips by frame count; deterministic, so every rank
    derives the same assignment without communicating."""
    bins: List[List[int]] = [[] for _ in range(world_size)]
    load = [0] * world_size
    for idx in sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i)):
        r = min(range(world_size), key=lambda k: (load[k], k))
        bins[r].append(idx)
        load[r] += int(lengths[idx])
    return bins


def gather_partials(local_rows: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """The single exchange step: all_gather of per-clip rows, padded to the largest shard."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    kmax = max(max(counts), 1)
    pad = torch.zeros((kmax, PARTIAL_COLS), dtype=torch.float64, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    bucket = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bucket, pad, group=group)
    return torch.cat([bucket[r][: counts[r]] for r in range(world)], dim=0)


def evaluate_clips(lift_clip: Callable, clips: Sequence[Clip], rf: int, device, flip: bool = False,
                   kps_left: Sequence[int] = (), kps_right: Sequence[int] = (),
                   rank: int = 0, world_size: int = 1, group=None, causal: bool = False,
                   joints_left: Optional[Sequence[int]] = None, joints_right: Optional[Sequence[int]] = None):
    """Evaluate `clips` (sharded over `world_size` ranks when > 1; `causal`: pad as main.py:85-89 does for
    CAUSAL models).  Every rank returns
    (per_action {name: (e1,e2,e3,ev,er) mm}, action-wise average, gathered partial rows)."""
    actions = sorted(set(c.action for c in clips))
    aid = {a: i for i, a in enumerate(actions)}
    shards = shard_clips([c.rays.shape[0] for c in clips], world_size)
    on_gpu = torch.device(device).type == "cuda"
    local = partial_rows([(idx, aid[clips[idx].action], clips[idx].rays.shape[0]) for idx in shards[rank]], device)
    for k, idx in enumerate(shards[rank]):
        c = clips[idx]
        pred = predict_clip(lift_clip, c, rf, device, flip, kps_left, kps_right, causal, joints_left, joints_right)
        cc = Clip(c.camera, c.rays, c.gt_norm, c.action, idx)
        if on_gpu and pred.is_cuda:
            clip_partials_hip(pred, cc, aid[c.action], out=local[k])
        else:
            local[k] = clip_partials(pred, cc, aid[c.action])
    if world_size > 1:
        allrows = gather_partials(local, [len(s) for s in shards], group)
    else:
        allrows = local
    per = reduce_partials(allrows)
    named = {actions[a]: v for a, v in per.items()}
    return named, action_average(per), allrows


def format_report(named: Dict[str, tuple], average: tuple) -> List[str]:
    """The lines Trainer.evaluate logs (lib/train_val/trainer.py:459-477): per action, then the action-wise
    averages rounded to 0.1 mm."""
    labels = ("Protocol #1 Error (MPJPE):  ", "Protocol #2 Error (P-MPJPE):", "Protocol #3 Error (N-MPJPE):",
              "Velocity    Error (MPJVE):  ", "Root        Error (MRPE):  ")
    lines = []
    for action, vals in named.items():
        lines.append("----" + action + "----")
        lines += ["%s %s mm" % (lab, v) for lab, v in zip(labels, vals)]
        lines.append("----------")
    heads = ("Protocol #1   (MPJPE)", "Protocol #2 (P-MPJPE)", "Protocol #3 (N-MPJPE)", "Velocity      (MPJVE)",
             "Root           (MRPE)")
    lines += ["%s action-wise average: %s mm" % (h, round(float(v), 1)) for h, v in zip(heads, average)]
    return lines
