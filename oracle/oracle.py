"""ctypes front end of oracle/libray3d_oracle.so - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(rules and parity status: oracle/ray3d_oracle.h).  The product package `ray3d_amd` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libray3d_oracle.so")


class _Tensor(C.Structure):
    _fields_ = [("key", C.c_char_p), ("data", C.c_void_p), ("rank", C.c_int),
                ("shape", C.c_int64 * 4)]


class _Config(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("kind", "num_joints", "in_features", "num_levels",
                                       "channels", "latent", "stage", "extrinsic_dim", "embed_dim", "causal", "dense")]


class _Camera(C.Structure):
    _fields_ = [("height", C.c_double), ("pitch", C.c_double),
                ("Rc2n", C.c_double * 9), ("Tc2n", C.c_double * 3),
                ("Rw2n", C.c_double * 9), ("Tw2n", C.c_double * 3),
                ("Rn2w", C.c_double * 9), ("Tn2w", C.c_double * 3)]


_TAP = C.CFUNCTYPE(None, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_int, C.c_void_p)
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with the committed Makefile (gcc only)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "ray3d_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libray3d_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.r3o_forward.restype = C.c_int
        _lib.r3o_forward.argtypes = [C.POINTER(_Config), C.POINTER(_Tensor), C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_int64, C.c_void_p, _TAP, C.c_void_p, C.c_int]
        _lib.r3o_last_error.restype = C.c_char_p
    return _lib


def _cfg_struct(cfg) -> _Config:
    return _Config(0 if cfg.kind == "pos" else 1, cfg.num_joints, cfg.in_features,
                   len(cfg.filter_widths), cfg.channels, cfg.latent, cfg.stage,
                   cfg.extrinsic_dim if cfg.camera_embedding else 0,
                   cfg.embed_dim if cfg.camera_embedding else 0, 1 if cfg.causal else 0,
                   1 if cfg.dense_convs else 0)


def forward(cfg, state: Dict[str, np.ndarray], x: np.ndarray, param: Optional[np.ndarray],
            taps: Optional[dict] = None, threads: int = 0) -> np.ndarray:
    """Reference-equivalent eval forward of RIEModel ('pos') / RIETrajectoryModel ('trj')."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    B = x.shape[0]
    assert x.shape == (B, cfg.receptive_field, cfg.num_joints, cfg.in_features), x.shape
    keep = []
    arr = (_Tensor * len(state))()
    n = 0
    for k, v in state.items():
        v = np.asarray(v)
        if v.dtype != np.float32:
            continue
        v = np.ascontiguousarray(v)
        keep.append(v)
        kb = k.encode()
        keep.append(kb)
        arr[n].key = kb
        arr[n].data = v.ctypes.data
        arr[n].rank = v.ndim
        for i, d in enumerate(v.shape):
            arr[n].shape[i] = d
        n += 1
    p = None
    if param is not None:
        p = np.ascontiguousarray(param, dtype=np.float32)
    out = np.empty((B, 1, cfg.num_joints if cfg.kind == "pos" else 1, 3), dtype=np.float32)

    def _tap(name, data, shape, rank, _user):
        shp = tuple(int(shape[i]) for i in range(rank))
        taps[name.decode()] = np.ctypeslib.as_array(data, shape=shp).copy()

    cb = _TAP(_tap) if taps is not None else C.cast(None, _TAP)
    c = _cfg_struct(cfg)
    rc = L.r3o_forward(C.byref(c), arr, n, x.ctypes.data, p.ctypes.data if p is not None else None,
                       B, out.ctypes.data, cb, None, threads)
    if rc != 0:
        raise RuntimeError("oracle: %s (code %d)" % (L.r3o_last_error().decode(), rc))
    return out


class Camera:
    """float64 restatement of the parts of CameraInfoPacket on the path (undistort=False)."""

    def __init__(self, K, R, t):
        self.K = np.ascontiguousarray(K, dtype=np.float64)
        R = np.ascontiguousarray(R, dtype=np.float64)
        t = np.ascontiguousarray(np.asarray(t, dtype=np.float64).reshape(3))
        self._c = _Camera()
        lib().r3o_camera_init(self.K.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p),
                              t.ctypes.data_as(C.c_void_p), C.byref(self._c))
        g = lambda a, s: np.array(a, dtype=np.float64).reshape(s)
        self.height, self.pitch = self._c.height, self._c.pitch
        self.Rc2n, self.Tc2n = g(self._c.Rc2n, (3, 3)), g(self._c.Tc2n, (3, 1))
        self.Rw2n, self.Tw2n = g(self._c.Rw2n, (3, 3)), g(self._c.Tw2n, (3, 1))
        self.Rn2w, self.Tn2w = g(self._c.Rn2w, (3, 3)), g(self._c.Tn2w, (3, 1))

    def _pts(self, fn, a, din, dout, with_k=True):
        a = np.ascontiguousarray(a, dtype=np.float64)
        n = a.size // din
        out = np.empty(a.shape[:-1] + (dout,), dtype=np.float64)
        args = [self.K.ctypes.data_as(C.c_void_p)] if with_k else []
        fn(*args, C.byref(self._c), a.ctypes.data_as(C.c_void_p), C.c_int64(n),
           out.ctypes.data_as(C.c_void_p))
        return out

    def rays_from_uv(self, uv):
        return self._pts(lib().r3o_rays_from_uv, uv, 2, 3)

    def uv_from_rays(self, rays):
        return self._pts(lib().r3o_uv_from_rays, rays, 3, 2)

    @staticmethod
    def transform(R, T, pts):
        R = np.ascontiguousarray(R, dtype=np.float64)
        T = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(3))
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        out = np.empty_like(pts)
        lib().r3o_transform(R.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p),
                            pts.ctypes.data_as(C.c_void_p), C.c_int64(pts.size // 3),
                            out.ctypes.data_as(C.c_void_p))
        return out

    def world2normalized(self, pts):
        return self.transform(self.Rw2n, self.Tw2n, pts)

    def normalized2world(self, pts):
        return self.transform(self.Rn2w, self.Tn2w, pts)


def _dist(fn, K, dist, uv):
    K = np.ascontiguousarray(K, dtype=np.float64)
    dist = np.ascontiguousarray(dist, dtype=np.float64).reshape(5)
    uv = np.ascontiguousarray(uv, dtype=np.float64)
    out = np.empty_like(uv)
    fn(K.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p),
       uv.ctypes.data_as(C.c_void_p), C.c_int64(uv.size // 2), out.ctypes.data_as(C.c_void_p))
    return out


def undistort_points(K, dist, uv):
    """PARITY UNPINNED (cv2.undistortPoints restated from its documentation)."""
    return _dist(lib().r3o_undistort_points, K, dist, uv)


def distort_points(K, dist, uv):
    return _dist(lib().r3o_distort_points, K, dist, uv)
