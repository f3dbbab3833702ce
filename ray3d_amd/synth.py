"""Deterministic, RNG-library-independent synthetic weights and inputs.

There is no network in the build/bench environment, so checkpoints and datasets are replaced by
tensors generated from a counter-based hash (splitmix64).  The same bytes come out on every
machine, numpy version and torch version, so the golden fixtures generated next to the reference
(tests/golden/make_golden.py) stay valid on the GPU box where the weights are *regenerated* from
the seed instead of being shipped (a RF-243 pos+trj state is ~200 MB).

Weight scales keep activation variance roughly constant through the LeakyReLU(0.2) stacks so that
an error in any layer is visible at the output, BatchNorm statistics are deliberately non-trivial,
and decoder outputs are of order one metre (the scale at which the 1e-4 abs tolerance is meant).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

from .spec import Entry, LiftConfig, state_entries

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def hash_uniform(name: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    """float64 array of i.i.d. U[0,1) values, a pure function of (name, seed, flat index)."""
    n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    base = np.uint64((_fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) + base) & _MASK
    bits = _splitmix64(_splitmix64(idx))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def _sym(name, shape, seed, bound):
    return (hash_uniform(name, shape, seed) * 2.0 - 1.0) * bound


_GAIN_LRELU = math.sqrt(2.0 / (1.0 + 0.2 ** 2))


def synth_tensor(e: Entry, seed: int = 0) -> np.ndarray:
    """One state_dict tensor (float32, or int64 for num_batches_tracked)."""
    if e.role in ("conv_w", "lin_w"):
        gain = _GAIN_LRELU if e.activated else 1.0
        bound = math.sqrt(3.0) * gain / math.sqrt(max(e.fan_in, 1))
        return _sym(e.key, e.shape, seed, bound).astype(np.float32)
    if e.role == "bias":
        return _sym(e.key, e.shape, seed, 0.1).astype(np.float32)
    if e.role == "bn_weight":
        return (0.8 + 0.4 * hash_uniform(e.key, e.shape, seed)).astype(np.float32)
    if e.role == "bn_bias":
        return _sym(e.key, e.shape, seed, 0.1).astype(np.float32)
    if e.role == "bn_mean":
        return _sym(e.key, e.shape, seed, 0.2).astype(np.float32)
    if e.role == "bn_var":
        return (0.5 + hash_uniform(e.key, e.shape, seed)).astype(np.float32)
    if e.role == "bn_count":
        return np.array(1000, dtype=np.int64)
    raise ValueError(e.role)


def synth_state(cfg: LiftConfig, seed: int = 0, out_scale: float = 1.0) -> Dict[str, np.ndarray]:
    """Full synthetic ``state_dict`` (numpy) for one network.

    ``out_scale`` multiplies the decoder's last linear layer so that outputs are metres-sized.
    The trajectory decoder additionally gets a bias of about (0, 0, 4) m - a person a few metres
    in front of the camera - so pos+trj looks like an absolute pose.
    """
    state = {}
    for e in state_entries(cfg, ):
        state[e.key] = synth_tensor(e, seed)
    for k in list(state):
        if k.startswith("Integration") and k.endswith("fc_2.weight"):
            state[k] = (state[k] * np.float32(out_scale)).astype(np.float32)
    if cfg.kind == "trj":
        b = state["Integration.fc_2.bias"].copy()
        b[2] += np.float32(4.0)
        state["Integration.fc_2.bias"] = b
    return state


def synth_rays(batch: int, cfg: LiftConfig, seed: int = 0, name: str = "rays") -> np.ndarray:
    """(batch, RF, J, F) float32 input that looks like ray-encoded keypoints of a moving person.

    F=3: (x/z, c*y+s, -s*y+c) around a slowly drifting skeleton (values like the reference's
    `get_cam_ray_given_uv` output for a camera with ~0.18 rad pitch); F=2: the first two only.
    """
    rf, j, f = cfg.receptive_field, cfg.num_joints, cfg.in_features
    base = _sym(name + ".base", (batch, 1, j, 2), seed, 0.35)
    drift = _sym(name + ".drift", (batch, 1, 1, 2), seed, 0.15)
    t = (np.arange(rf, dtype=np.float64) / max(rf - 1, 1) - 0.5).reshape(1, rf, 1, 1)
    jitter = _sym(name + ".jit", (batch, rf, j, 2), seed, 0.02)
    xy = base + drift * t + jitter
    if f == 2:
        return xy.astype(np.float32)
    c, s = math.cos(0.18404), math.sin(0.18404)
    out = np.empty((batch, rf, j, 3), dtype=np.float64)
    out[..., 0] = xy[..., 0]
    out[..., 1] = c * xy[..., 1] + s
    out[..., 2] = -s * xy[..., 1] + c
    return out.astype(np.float32)


def synth_param(batch: int, seed: int = 0, name: str = "param", vary: bool = True) -> np.ndarray:
    """(batch, 2) float32 [camera height (m), pitch (rad)] (lib/train_val/trainer.py:297)."""
    p = np.empty((batch, 2), dtype=np.float64)
    if vary:
        p[:, 0] = 1.2 + 0.8 * hash_uniform(name + ".h", (batch,), seed)
        p[:, 1] = -0.1 + 0.5 * hash_uniform(name + ".p", (batch,), seed)
    else:
        p[:, 0] = 1.4812
        p[:, 1] = 0.18404
    return p.astype(np.float32)
