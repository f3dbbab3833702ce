"""numpy restatement of lib/loss/loss.py - TEST INFRASTRUCTURE ONLY (see ray3d_oracle.h).

Inputs are (..., J, 3) arrays; every function returns a python float.  Checked against
tests/golden/losses.npz (values produced by the reference's own functions).
"""
import numpy as np


def mpjpe(pred, target):
    """loss.py:12-18"""
    return float(np.mean(np.linalg.norm(pred - target, axis=-1)))


def n_mpjpe(pred, target):
    """loss.py:72-82 (scale-only alignment; expects (N, T, J, 3))"""
    norm_p = np.mean(np.sum(pred ** 2, axis=3, keepdims=True), axis=2, keepdims=True)
    norm_t = np.mean(np.sum(target * pred, axis=3, keepdims=True), axis=2, keepdims=True)
    return mpjpe(norm_t / norm_p * pred, target)


def p_mpjpe(pred, target):
    """loss.py:30-69 (Procrustes; expects (N, J, 3))"""
    muX, muY = target.mean(axis=1, keepdims=True), pred.mean(axis=1, keepdims=True)
    X0, Y0 = target - muX, pred - muY
    nX = np.sqrt((X0 ** 2).sum(axis=(1, 2), keepdims=True))
    nY = np.sqrt((Y0 ** 2).sum(axis=(1, 2), keepdims=True))
    X0, Y0 = X0 / nX, Y0 / nY
    H = X0.transpose(0, 2, 1) @ Y0
    U, s, Vt = np.linalg.svd(H)
    V = Vt.transpose(0, 2, 1)
    R = V @ U.transpose(0, 2, 1)
    sign = np.sign(np.linalg.det(R))[:, None]
    V[:, :, -1] *= sign
    s[:, -1] *= sign.flatten()
    R = V @ U.transpose(0, 2, 1)
    tr = s.sum(axis=1, keepdims=True)[:, :, None]
    a = tr * nX / nY
    t = muX - a * (muY @ R)
    return float(np.mean(np.linalg.norm(a * (pred @ R) + t - target, axis=-1)))


def mean_velocity_error(pred, target):
    """loss.py:95-104 (expects (N, J, 3), velocity along axis 0)"""
    return float(np.mean(np.linalg.norm(np.diff(pred, axis=0) - np.diff(target, axis=0), axis=-1)))
