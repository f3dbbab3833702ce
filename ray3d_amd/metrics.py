"""Pose-error metrics of the evaluation loop, in torch (float64, any device).

Counterparts of lib/loss/loss.py: mpjpe (:12-18), n_mpjpe (:72-82), p_mpjpe (:30-69, NumPy SVD in
the reference, batched torch.linalg.svd here so it can stay on the GPU), mean_velocity_error
(:95-104).  All take (..., J, 3) tensors and return a 0-d tensor.

HOST-SIDE ONLY: the product path computes these on the device in float64 (r3d_clip_metrics, csrc/r3d_metrics.hip), which
is what `evaluate.clip_partials` calls for every CUDA tensor.  This module serves CPU tensors - the host-logic tests that
drive the evaluation loop with a stand-in lifter (tests/test_host.py) and callers without a GPU tensor in hand; it is
never a fallback for the HIP kernels.
"""
from __future__ import annotations

import torch


def mpjpe(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    assert pred.shape == target.shape
    return torch.linalg.vector_norm(pred - target, dim=-1).mean()


def n_mpjpe(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Scale-aligned MPJPE; per-frame scale = <target,pred> / <pred,pred> over joints."""
    assert pred.shape == target.shape
    num = (target * pred).sum(dim=-1, keepdim=True).mean(dim=-2, keepdim=True)
    den = (pred * pred).sum(dim=-1, keepdim=True).mean(dim=-2, keepdim=True)
    return mpjpe(num / den * pred, target)


def p_mpjpe(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Procrustes-aligned MPJPE (similarity transform per frame)."""
    assert pred.shape == target.shape
    pred = pred.reshape(-1, pred.shape[-2], 3)
    target = target.reshape(-1, target.shape[-2], 3)
    mu_t, mu_p = target.mean(dim=1, keepdim=True), pred.mean(dim=1, keepdim=True)
    t0, p0 = target - mu_t, pred - mu_p
    nt = torch.sqrt((t0 ** 2).sum(dim=(1, 2), keepdim=True))
    np_ = torch.sqrt((p0 ** 2).sum(dim=(1, 2), keepdim=True))
    t0, p0 = t0 / nt, p0 / np_
    H = t0.transpose(1, 2) @ p0
    U, s, Vt = torch.linalg.svd(H)
    V = Vt.transpose(1, 2)
    R = V @ U.transpose(1, 2)
    sign = torch.sign(torch.linalg.det(R)).unsqueeze(1)      # undo reflections
    V = torch.cat([V[:, :, :-1], V[:, :, -1:] * sign.unsqueeze(2)], dim=2)
    s = torch.cat([s[:, :-1], s[:, -1:] * sign], dim=1)
    R = V @ U.transpose(1, 2)
    a = s.sum(dim=1, keepdim=True).unsqueeze(2) * nt / np_
    t = mu_t - a * (mu_p @ R)
    return torch.linalg.vector_norm(a * (pred @ R) + t - target, dim=-1).mean()


def mean_velocity_error(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Mean norm of the first temporal difference of the error; frames along dim 0."""
    assert pred.shape == target.shape
    if pred.shape[0] < 2:
        return torch.full((), float("nan"), dtype=pred.dtype, device=pred.device)
    return torch.linalg.vector_norm(torch.diff(pred, dim=0) - torch.diff(target, dim=0), dim=-1).mean()
