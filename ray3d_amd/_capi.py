"""ctypes binding of libray3d_hip.so (C ABI: include/ray3d_hip.h).

There is deliberately no fallback: if the shared library is missing or a call fails, this module
raises.  It never imports anything from ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libray3d_hip.so")
# The hooks build (the same sources with -DR3D_TEST_HOOKS): the r3d_debug_* exports and the development switches read from
# the environment.  tests/ and tools/ select it with use_hooks(True); nothing in this package does.
# (R3D_HOOKS_LIB: another build of the hooks library, e.g. libray3d_hip_san.so - `make -C ray3d_amd/csrc san`: host objects under ASan / UBSan)
HOOKS_LIB_PATH = os.environ.get("R3D_HOOKS_LIB") or os.path.join(_HERE, "libray3d_hip_hooks.so")

R3D_KIND_POS, R3D_KIND_TRJ = 0, 1
R3D_INPUT_RAYS, R3D_INPUT_UV = 0, 1
R3D_ERR_ABORTED = -7
R3D_OPT_STAGED, R3D_OPT_SPIN_TIMEOUT_MS, R3D_OPT_CU_LIMIT, R3D_OPT_LANES = 1, 2, 3, 4

# every symbol include/ray3d_hip.h declares (tests check the library exports exactly these)
EXPORTS = (
    "r3d_create", "r3d_destroy", "r3d_num_weights", "r3d_weight_key", "r3d_weight_shape",
    "r3d_set_weight", "r3d_finalize", "r3d_workspace_bytes", "r3d_forward", "r3d_forward_pair",
    "r3d_profile_enable", "r3d_profile_read", "r3d_clip_metrics", "r3d_last_error", "r3d_version",
    "r3d_prepare", "r3d_release", "r3d_abi_version", "r3d_precision", "r3d_status", "r3d_set_option", "r3d_last_clock",
    "r3d_lane_stream", "r3d_lanes_join",
)
HOOK_EXPORTS = ("r3d_debug_schedule_check", "r3d_debug_plan_check", "r3d_debug_forward_check")   # libray3d_hip_hooks.so only
ABI_VERSION = 6                                                          # R3D_ABI_VERSION of the header this binding follows
METRIC_NAMES = ("mpjpe", "p_mpjpe", "n_mpjpe", "velocity", "root")     # R3D_METRIC_* order
METRIC_OUT_DOUBLES = 5 * (1 + 128)                                      # R3D_METRIC_OUT_DOUBLES


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("struct_size", "kind", "num_joints", "in_features", "num_levels",
                                         "channels", "latent", "stage", "extrinsic_dim",
                                         "embed_dim", "causal", "dense", "bf16x3")]


class Input(C.Structure):
    _fields_ = [("mode", C.c_int32), ("x_dev", C.c_void_p), ("window_stride", C.c_int64),
                ("param_dev", C.c_void_p), ("param_stride", C.c_int64),
                ("cam_dev", C.c_void_p), ("cam_stride", C.c_int64)]


class LaunchRecord(C.Structure):
    _fields_ = [("kernel", C.c_char * 48), ("stage", C.c_int32), ("blocks", C.c_int32),
                ("ms", C.c_float), ("flops", C.c_double), ("bytes", C.c_double)]


class Ray3DHipError(RuntimeError):
    pass


_libs = {}
_hooks = os.environ.get("R3D_USE_HOOKS_LIB", "0") not in ("", "0")   # (tools/: A/B runs of bench.py on the hooks build)


_live = {}      # library path -> number of live Handles it created


def use_hooks(on: bool) -> None:
    """Tests / tools: make load() return libray3d_hip_hooks.so (True) or the product library (False, the default).  Handles
    belong to the library that created them: switch before building modules, and do not carry them across a switch.
    The two libraries are two copies of the same code with their OWN process-wide state - in particular the ordering of
    single-launch forwards of different streams (each needs every CU): forwards issued through both copies are not ordered
    against each other.  So when handles of the library being left are still alive, the switch first waits for the device -
    nothing of theirs is in flight when the other copy launches.  Do not run forwards of both copies concurrently."""
    global _hooks
    if bool(on) != _hooks and _live.get(HOOKS_LIB_PATH if _hooks else LIB_PATH, 0) > 0:
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except ImportError:
            pass
    _hooks = bool(on)


def load():
    """Load libray3d_hip.so (or, after use_hooks(True), its hooks build); raise (never fall back) when it is not there."""
    path = HOOKS_LIB_PATH if _hooks else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise Ray3DHipError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
    lib = C.CDLL(path)
    vp, i64p = C.c_void_p, C.POINTER(C.c_int64)
    lib.r3d_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.r3d_destroy.argtypes = [vp]
    lib.r3d_num_weights.argtypes = [vp]
    lib.r3d_weight_key.argtypes = [vp, C.c_int]
    lib.r3d_weight_key.restype = C.c_char_p
    lib.r3d_weight_shape.argtypes = [vp, C.c_int, i64p, C.POINTER(C.c_int)]
    lib.r3d_set_weight.argtypes = [vp, C.c_char_p, vp, i64p, C.c_int]
    lib.r3d_finalize.argtypes = [vp]
    lib.r3d_workspace_bytes.argtypes = [vp, vp, C.c_int64]
    lib.r3d_workspace_bytes.restype = C.c_size_t
    lib.r3d_forward.argtypes = [vp, C.POINTER(Input), C.c_int64, vp, vp, C.c_size_t, vp]
    lib.r3d_forward_pair.argtypes = [vp, vp, C.POINTER(Input), C.c_int64, vp, vp, vp, C.c_size_t, vp]
    lib.r3d_prepare.argtypes = [vp, vp, C.c_int64]
    lib.r3d_release.argtypes = [vp, vp, C.c_int64]
    lib.r3d_precision.argtypes = [vp]
    lib.r3d_status.argtypes = [vp, vp]
    lib.r3d_set_option.argtypes = [vp, C.c_int32, C.c_int64]
    lib.r3d_last_clock.argtypes = [vp, vp, C.POINTER(C.c_double)]
    lib.r3d_lane_stream.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    lib.r3d_lanes_join.argtypes = [vp, vp]
    lib.r3d_profile_enable.argtypes = [vp, C.c_int]
    lib.r3d_profile_read.argtypes = [vp, C.POINTER(LaunchRecord), C.c_int]
    lib.r3d_clip_metrics.argtypes = [vp, vp, C.c_int64, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), vp, vp]
    lib.r3d_last_error.restype = C.c_char_p
    lib.r3d_version.restype = C.c_char_p
    for name in EXPORTS + (HOOK_EXPORTS if _hooks else ()):
        fn = getattr(lib, name)
        if fn.restype is C.c_int or fn.restype is None:
            fn.restype = C.c_int
    if lib.r3d_abi_version() != ABI_VERSION:
        raise Ray3DHipError("%s has ABI version %d, this binding was written for %d: rebuild the library"
                            % (path, lib.r3d_abi_version(), ABI_VERSION))
    _libs[path] = lib
    return lib


def check(rc: int, what: str):
    if rc < 0:
        raise Ray3DHipError("%s failed (%d): %s" % (what, rc, load().r3d_last_error().decode()))
    return rc


class Handle:
    """Owns one r3d_model*."""

    def __init__(self, cfg):
        lib = self._lib = load()          # (a handle stays with the library that created it: see use_hooks)
        c = Config(C.sizeof(Config), R3D_KIND_POS if cfg.kind == "pos" else R3D_KIND_TRJ, cfg.num_joints,
                   cfg.in_features, len(cfg.filter_widths), cfg.channels, cfg.latent, cfg.stage,
                   cfg.extrinsic_dim if cfg.camera_embedding else 0,
                   cfg.embed_dim if cfg.camera_embedding else 0, 1 if cfg.causal else 0,
                   1 if cfg.dense_convs else 0, 1 if getattr(cfg, "bf16x3", False) else 0)
        self.ptr = C.c_void_p()
        check(lib.r3d_create(C.byref(c), C.byref(self.ptr)), "r3d_create")
        self._lib_path = HOOKS_LIB_PATH if lib is _libs.get(HOOKS_LIB_PATH) else LIB_PATH
        _live[self._lib_path] = _live.get(self._lib_path, 0) + 1

    def keys(self) -> List[str]:
        lib = self._lib
        return [lib.r3d_weight_key(self.ptr, i).decode() for i in range(lib.r3d_num_weights(self.ptr))]

    def shape(self, index: int):
        shp = (C.c_int64 * 4)()
        rank = C.c_int()
        check(self._lib.r3d_weight_shape(self.ptr, index, shp, C.byref(rank)), "r3d_weight_shape")
        return tuple(int(shp[i]) for i in range(rank.value))

    def set_weight(self, key: str, array):
        """array: C-contiguous float32 numpy array in torch layout."""
        shp = (C.c_int64 * 4)(*([int(d) for d in array.shape] + [1] * (4 - array.ndim)))
        check(self._lib.r3d_set_weight(self.ptr, key.encode(), array.ctypes.data_as(C.c_void_p), shp,
                                    array.ndim), "r3d_set_weight(%s)" % key)

    def finalize(self):
        check(self._lib.r3d_finalize(self.ptr), "r3d_finalize")

    def precision(self) -> str:
        """'f32' or 'bf16x3': what the handle's large GEMMs run in (r3d_config.bf16x3 or the R3D_BF16X3 override)."""
        return "bf16x3" if check(self._lib.r3d_precision(self.ptr), "r3d_precision") == 1 else "f32"

    def set_option(self, option: int, value: int):
        check(self._lib.r3d_set_option(self.ptr, option, value), "r3d_set_option")

    def lane_stream(self, lane: int) -> int:
        """r3d_lane_stream: the hipStream_t of lane `lane` of a handle with R3D_OPT_LANES (the library owns it)."""
        st = C.c_void_p()
        check(self._lib.r3d_lane_stream(self.ptr, lane, C.byref(st)), "r3d_lane_stream")
        return int(st.value)

    def lanes_join(self, stream: int):
        """r3d_lanes_join: `stream` waits for the forwards the library relayed to lanes from other streams."""
        check(self._lib.r3d_lanes_join(self.ptr, stream), "r3d_lanes_join")

    def status(self, stream: int) -> bool:
        """r3d_status: synchronises `stream`; True when every forward of this handle since the last call finished, False
        when one gave up waiting for its own tiles (outputs NaN; the flag is cleared)."""
        rc = self._lib.r3d_status(self.ptr, stream)
        if rc == R3D_ERR_ABORTED:
            return False
        check(rc, "r3d_status")
        return True

    def last_clock_ghz(self, stream: int) -> float:
        """r3d_last_clock: the shader clock the handle's last single-launch forward ran at (0.0: none / level by level)."""
        ghz = C.c_double(0.0)
        check(self._lib.r3d_last_clock(self.ptr, stream, C.byref(ghz)), "r3d_last_clock")
        return float(ghz.value)

    def profile_enable(self, on: bool):
        check(self._lib.r3d_profile_enable(self.ptr, 1 if on else 0), "r3d_profile_enable")

    def profile_read(self):
        cap = 128
        recs = (LaunchRecord * cap)()
        n = check(self._lib.r3d_profile_read(self.ptr, recs, cap), "r3d_profile_read")
        return [dict(kernel=recs[i].kernel.decode(), stage=recs[i].stage, blocks=recs[i].blocks,
                     ms=recs[i].ms, flops=recs[i].flops, bytes=recs[i].bytes) for i in range(min(n, cap))]

    def close(self):
        if getattr(self, "ptr", None) and self.ptr.value and getattr(self, "_lib", None) is not None:
            self._lib.r3d_destroy(self.ptr)
            self.ptr = C.c_void_p()
            _live[self._lib_path] = _live.get(self._lib_path, 1) - 1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _lib_of(*handles):
    """The library the given handles belong to (the selected one when there is none)."""
    for h in handles:
        if h is not None:
            return h._lib
    return load()


def workspace_bytes(pos: Optional[Handle], trj: Optional[Handle], batch: int) -> int:
    return int(_lib_of(pos, trj).r3d_workspace_bytes(pos.ptr if pos else None, trj.ptr if trj else None, batch))


def prepare(pos: Optional[Handle], trj: Optional[Handle], batch: int):
    """r3d_prepare: plan + tile schedule of this batch size, uploaded (outside of any stream capture)."""
    check(_lib_of(pos, trj).r3d_prepare(pos.ptr if pos else None, trj.ptr if trj else None, batch), "r3d_prepare")


def release(pos: Optional[Handle], trj: Optional[Handle], batch: int):
    """r3d_release: un-pin a batch size named in prepare() (after the hipGraph that captured it is gone)."""
    check(_lib_of(pos, trj).r3d_release(pos.ptr if pos else None, trj.ptr if trj else None, batch), "r3d_release")


def make_input(mode, x_ptr, window_stride, param_ptr, param_stride, cam_ptr=None, cam_stride=0) -> Input:
    return Input(mode, x_ptr, window_stride, param_ptr, param_stride, cam_ptr, cam_stride)


def forward(handle: Handle, inp: Input, batch: int, out_ptr: int, ws_ptr: int, ws_bytes: int, stream: int):
    check(handle._lib.r3d_forward(handle.ptr, C.byref(inp), batch, out_ptr, ws_ptr, ws_bytes, stream),
          "r3d_forward")


def clip_metrics(pred_ptr: int, gt_ptr: int, n_frames: int, num_joints: int, rn2w, tn2w, out_ptr: int, stream: int):
    """r3d_clip_metrics: rn2w (3,3) / tn2w (3,) float64 array-likes on the host; the rest device pointers."""
    r = (C.c_double * 9)(*[float(v) for row in rn2w for v in row])
    t = (C.c_double * 3)(*[float(v) for v in tn2w])
    check(load().r3d_clip_metrics(pred_ptr, gt_ptr, n_frames, num_joints, r, t, out_ptr, stream), "r3d_clip_metrics")


def forward_pair(pos: Handle, trj: Handle, inp: Input, batch: int, out_ptr: int,
                 out_trj_ptr: Optional[int], ws_ptr: int, ws_bytes: int, stream: int):
    check(pos._lib.r3d_forward_pair(pos.ptr, trj.ptr, C.byref(inp), batch, out_ptr, out_trj_ptr,
                                  ws_ptr, ws_bytes, stream), "r3d_forward_pair")
