"""nn.Module front end that keeps the reference's module contract but runs on libray3d_hip.so.

Drop-in seam:

* ``Model(model_config, data_config, is_train)`` with ``get_pos_model()`` / ``get_trj_model()``
  - lib/model/__init__.py:5-62
* ``pos_model(inputs_2d (B,RF,J,F), inputs_param (B,E)) -> (B,1,J,3)``  - lib/model/rie.py:284-434
* ``trj_model(...) -> (B,1,1,3)``                                        - lib/model/rie.py:518-559
* ``state_dict`` / ``load_state_dict(strict=True)`` with the reference's key names and shapes
  (lib/model/rie.py constructors), ``receptive_field()``, ``eval()/train()``.

The modules only *hold* parameters (so checkpoints and optimisers see the usual tensors); the
arithmetic happens in the HIP library.  Forward is inference-only (eval-mode BatchNorm, dropout
off) - calling it in training mode, on a CPU tensor or without the built library raises; there
is no PyTorch or CPU fallback.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import os
import torch
import torch.nn as nn

from . import _capi
from .spec import LiftConfig, config_from_dicts, state_entries


class _Node(nn.Module):
    """Anonymous container; the tree of these mirrors the reference's attribute paths."""


def _grow_tree(root: nn.Module, cfg: LiftConfig) -> None:
    for e in state_entries(cfg):
        *path, leaf = e.key.split(".")
        node = root
        for part in path:
            if part not in node._modules:
                node.add_module(part, _Node())
            node = node._modules[part]
        if e.role == "bn_count":
            node.register_buffer(leaf, torch.tensor(0, dtype=torch.long))
        elif e.is_buffer:
            fill = 1.0 if e.role == "bn_var" else 0.0
            node.register_buffer(leaf, torch.full(e.shape, fill, dtype=torch.float32))
        else:
            t = torch.empty(e.shape, dtype=torch.float32)
            if e.role in ("conv_w", "lin_w"):        # nn.Conv1d / nn.Linear default init
                nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            elif e.role == "bias":
                b = 1.0 / math.sqrt(max(e.fan_in, 1))
                nn.init.uniform_(t, -b, b)
            elif e.role == "bn_weight":
                t.fill_(1.0)
            else:
                t.zero_()
            node.register_parameter(leaf, nn.Parameter(t))


class _Workspace:
    """Grow-only scratch HBM per device, handed to the C ABI (the caller owns device buffers)."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(int(nbytes * 1.0) + 256, dtype=torch.uint8, device=device)
        return self.buf


class LiftModule(nn.Module):
    """Common machinery of RIEModel / RIETrajectoryModel."""

    KIND = "pos"

    def __init__(self, cfg: LiftConfig):
        super().__init__()
        self.cfg = cfg
        self.num_joints_in = cfg.num_joints
        self.num_joints_out = cfg.num_joints
        self.in_features = cfg.in_features
        self.latten_features = cfg.latent
        self.stage = cfg.stage
        self.camera_embedding = cfg.camera_embedding
        self.extrinsic_dim = cfg.extrinsic_dim
        self.embedd_dim = cfg.embed_dim
        self.pad = (cfg.receptive_field - 1) // 2
        _grow_tree(self, cfg)
        self._handle: Optional[_capi.Handle] = None
        self._synced = False
        self._staged = False
        self._spin_timeout_ms: Optional[int] = None
        self._cu_limit = 0
        self._ws = _Workspace()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate())

    # ---- reference module surface -------------------------------------------------------
    def receptive_field(self) -> int:
        """lib/model/rie.py:278-282 / :512-516"""
        return self.cfg.receptive_field

    def set_bn_momentum(self, momentum):   # rie.py:255-260 (training only; no effect on eval)
        self._bn_momentum = momentum

    def set_training_status(self, is_train):   # rie.py:262-268
        self.is_train = is_train

    def set_augment(self, augment):   # rie.py:270-276
        self.augment = augment

    # ---- weight sync ---------------------------------------------------------------------
    def _invalidate(self):
        self._synced = False

    def _apply(self, fn, *args, **kwargs):   # .cuda() / .to() / .float() move the tensors
        self._invalidate()
        return super()._apply(fn, *args, **kwargs)

    def refresh_weights(self):
        """Call after editing parameters in place (optimizer step, manual surgery)."""
        self._invalidate()

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        # nn.DataParallel checkpoints (lib/model/__init__.py:52) carry a 'module.' prefix
        if any(k.startswith("module.") for k in state_dict):
            state_dict = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def set_staged(self, on: bool = True):
        """r3d_set_option(R3D_OPT_STAGED): this module's forwards run as one launch per level of the network instead of one
        persistent launch - the form that needs no co-residency of its workgroups (a GPU shared with other processes)."""
        self._staged = bool(on)
        if self._handle is not None:
            self._handle.set_option(_capi.R3D_OPT_STAGED, 1 if self._staged else 0)

    def set_spin_timeout_ms(self, ms: int):
        """r3d_set_option(R3D_OPT_SPIN_TIMEOUT_MS): how long a tile of the single-launch forward waits for its producers
        before the forward gives up (default 1000)."""
        self._spin_timeout_ms = int(ms)
        if self._handle is not None:
            self._handle.set_option(_capi.R3D_OPT_SPIN_TIMEOUT_MS, self._spin_timeout_ms)

    def set_cu_limit(self, n: int):
        """r3d_set_option(R3D_OPT_CU_LIMIT): this module's forwards run on a CU-masked stream of `n` CUs (0: the whole
        device) - see :func:`ray3d_amd.masked_stream` and Ray3DLifter.set_cu_limit."""
        self._cu_limit = int(n)
        if self._handle is not None:
            self._handle.set_option(_capi.R3D_OPT_CU_LIMIT, self._cu_limit)

    def check_status(self, device=None) -> None:
        """Synchronises the current stream of `device` and raises when a forward of this module since the last check
        gave up waiting for its own tiles (r3d_status: its outputs are NaN).  The reference's seam reports errors as
        Python exceptions; a forward is asynchronous, so this is where a device-side failure surfaces."""
        dev = torch.device(device) if device is not None else getattr(self, "_device", None)
        if self._handle is None or dev is None:
            return
        with torch.cuda.device(dev):
            if not self._handle.status(torch.cuda.current_stream(dev).cuda_stream):
                raise _capi.Ray3DHipError("forward aborted: " + self._handle._lib.r3d_last_error().decode())

    def handle(self, device: torch.device) -> _capi.Handle:
        """The finalized C handle for `device`, re-uploading weights when they changed."""
        if self._handle is None:
            self._handle = _capi.Handle(self.cfg)
            if self._staged:
                self._handle.set_option(_capi.R3D_OPT_STAGED, 1)
            if self._spin_timeout_ms is not None:
                self._handle.set_option(_capi.R3D_OPT_SPIN_TIMEOUT_MS, self._spin_timeout_ms)
            if self._cu_limit:
                self._handle.set_option(_capi.R3D_OPT_CU_LIMIT, self._cu_limit)
        if not self._synced or getattr(self, "_device", None) != device:
            sd = self.state_dict()
            for key in self._handle.keys():
                arr = sd[key].detach().to("cpu", torch.float32).contiguous().numpy()
                self._handle.set_weight(key, np.ascontiguousarray(arr))
            with torch.cuda.device(device):
                self._handle.finalize()
            self._synced, self._device = True, device
        return self._handle

    # ---- forward ---------------------------------------------------------------------------
    def _check_inputs(self, x: torch.Tensor, param: Optional[torch.Tensor]):
        assert len(x.shape) == 4                              # rie.py:285
        assert x.shape[-2] == self.num_joints_in              # rie.py:286
        assert x.shape[-1] == self.in_features                # rie.py:287
        if x.shape[1] != self.cfg.receptive_field:
            raise RuntimeError("expected %d-frame windows (1 output frame per window, quirk F3), got %d"
                               % (self.cfg.receptive_field, x.shape[1]))
        if self.training:
            raise RuntimeError("ray3d_amd modules are inference-only: call .eval() first "
                               "(training-mode BatchNorm/Dropout are not implemented)")
        if not x.is_cuda:
            raise RuntimeError("ray3d_amd runs on an AMD GPU only (got a %s tensor); there is no "
                               "CPU fallback" % x.device)
        if self.camera_embedding:
            if param is None or param.shape != (x.shape[0], self.extrinsic_dim):
                raise RuntimeError("inputs_param must have shape (%d, %d)" % (x.shape[0], self.extrinsic_dim))

    def forward(self, x: torch.Tensor, param: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._check_inputs(x, param)
        dev = x.device
        x = x.detach().to(torch.float32).contiguous()
        p = param.detach().to(dev, torch.float32).contiguous() if self.camera_embedding else None
        B, J = x.shape[0], self.num_joints_in
        out = torch.empty((B, 1, J if self.KIND == "pos" else 1, 3), dtype=torch.float32, device=dev)
        h = self.handle(dev)
        nbytes = _capi.workspace_bytes(h if self.KIND == "pos" else None, h if self.KIND == "trj" else None, B)
        ws = self._ws.get(nbytes, dev)
        inp = _capi.make_input(_capi.R3D_INPUT_RAYS, x.data_ptr(), self.cfg.receptive_field,
                               p.data_ptr() if p is not None else None, self.extrinsic_dim)
        with torch.cuda.device(dev):
            _capi.forward(h, inp, B, out.data_ptr(), ws.data_ptr(), ws.numel(),
                          torch.cuda.current_stream(dev).cuda_stream)
        return out


class RIEModel(LiftModule):
    """Pose network, lib/model/rie.py:172-434."""
    KIND = "pos"


class RIETrajectoryModel(LiftModule):
    """Root-trajectory network, lib/model/rie.py:437-559."""
    KIND = "trj"


class _SingleDeviceParallel(nn.Module):
    """What the reference's ``nn.DataParallel(model).cuda()`` (lib/model/__init__.py:51-53) looks
    like from the outside - a ``.module`` attribute and 'module.'-prefixed state_dict keys - without
    the per-call replicate/scatter/gather: multi-GPU here is one process per GPU (ray3d_amd.dist)."""

    def __init__(self, module: LiftModule):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def receptive_field(self):
        return self.module.receptive_field()

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        if state_dict and not any(k.startswith("module.") for k in state_dict):
            state_dict = {"module." + k: v for k, v in state_dict.items()}
        return super().load_state_dict(state_dict, strict=strict, **kw)


class Model(object):
    """Factory with the reference's surface (lib/model/__init__.py:5-62)."""

    def __init__(self, model_config: dict, data_config: Optional[dict] = None, is_train: bool = True):
        pos_model = RIEModel(config_from_dicts(model_config, "pos"))
        trj_model = RIETrajectoryModel(config_from_dicts(model_config, "trj")) \
            if model_config["TRAJECTORY_MODEL"] else None
        for m in (pos_model, trj_model):
            if m is not None:
                m.is_train = is_train
                m.train(is_train)
        if torch.cuda.is_available():
            pos_model = _SingleDeviceParallel(pos_model).cuda()
            trj_model = _SingleDeviceParallel(trj_model).cuda() if trj_model is not None else None
        self.pos_model = pos_model
        self.trj_model = trj_model

    def get_pos_model(self):
        return self.pos_model

    def get_trj_model(self):
        return self.trj_model


def _unwrap(m) -> LiftModule:
    return m.module if isinstance(m, _SingleDeviceParallel) else m


class Ray3DLifter(nn.Module):
    """pos + trj in one pass: ``lifter(x, param) == pos_model(x, param) + trj_model(x, param)``
    (lib/train_val/trainer.py:337,346,353) with the prologue shared and every level of both
    networks batched into common launches."""

    def __init__(self, pos_model, trj_model):
        super().__init__()
        self.pos = _unwrap(pos_model)
        self.trj = _unwrap(trj_model)
        if not isinstance(self.pos, RIEModel) or not isinstance(self.trj, RIETrajectoryModel):
            raise TypeError("Ray3DLifter(pos_model: RIEModel, trj_model: RIETrajectoryModel)")
        self._ws = _Workspace()
        self._side_ws: list = []
        self._side_streams: list = []
        self._prepared: set = set()

    def receptive_field(self) -> int:
        return self.pos.receptive_field()

    def set_staged(self, on: bool = True):
        """Both networks level by level (R3D_OPT_STAGED) instead of one persistent launch: see LiftModule.set_staged."""
        self.pos.set_staged(on)
        self.trj.set_staged(on)

    def set_spin_timeout_ms(self, ms: int):
        self.pos.set_spin_timeout_ms(ms)
        self.trj.set_spin_timeout_ms(ms)

    def set_cu_limit(self, n: int):
        """The pair's forwards run on a CU-masked stream that can use `n` CUs (R3D_OPT_CU_LIMIT; 0: the whole device): the
        single persistent launch then uses at most `n` workgroups and is not ordered against forwards of other streams.
        For two lifters side by side on disjoint halves of the chip (ray3d_amd.masked_stream) - an experiment reported
        by bench.py --half-chip-streams, not the default path."""
        self.release_prepared()
        self.pos.set_cu_limit(n)
        self.trj.set_cu_limit(n)

    def set_lanes(self, n: int, device=None):
        """R3D_OPT_LANES: `n` (2 or 4; 0: off) library-owned CU-masked streams per pair - lane k: the CUs c of every XCD with
        c % n == k - each with its own tile schedules and control regions, ONE packed weight image for all of them.  For
        callers with independent batches in flight (the clip evaluation: 240 clips, lib/train_val/trainer.py:295-353): n
        forwards share the chip side by side.  Use :meth:`lane` to run a clip's forward AND what consumes its poses on a
        lane's stream; :meth:`join_lanes` makes the current stream wait for all lanes.  A forward issued on any other
        stream is relayed to the next lane round-robin (its result is ordered behind :meth:`join_lanes`, not behind the call).
        Both handles are finalised on `device` first (the lanes live on the handles' device)."""
        dev = torch.device(device) if device is not None else (getattr(self.pos, "_device", None) or torch.device("cuda", torch.cuda.current_device()))
        n = int(n)
        n = 0 if n <= 1 else n
        hp, ht = self.pos.handle(dev), self.trj.handle(dev)
        self.release_prepared()
        if getattr(self, "_lane_keep", None):
            torch.cuda.synchronize(dev)          # (their events live on the lanes' streams, which the option change destroys)
            self._lane_keep = []
        with torch.cuda.device(dev):
            hp.set_option(_capi.R3D_OPT_LANES, n)
            ht.set_option(_capi.R3D_OPT_LANES, n)
        self._lanes = n
        self._lane_dev = dev
        self._lane_streams = [torch.cuda.ExternalStream(hp.lane_stream(k), device=dev) for k in range(n)]
        self._lane_ws = [_Workspace() for _ in range(n)]
        self._lane_rr = 0
        self._lane_pending = [set() for _ in range(n)]
        self._lane_keep = []                 # (event, tensors) of relayed forwards still in flight: see _run

    def num_lanes(self) -> int:
        return getattr(self, "_lanes", 0)

    def lane_stream(self, k: int):
        """Lane k's stream as a torch.cuda.ExternalStream (library-owned: valid while the pos handle lives)."""
        return self._lane_streams[k]

    def lane(self, k: Optional[int] = None):
        """Context manager: lane k's stream (the next lane round-robin when k is None) as the current stream, behind everything
        the caller's stream holds so far.  Forwards and whatever consumes their outputs inside the block run on that lane; the
        caller's stream sees the results after :meth:`join_lanes`."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            if not self.num_lanes():
                yield None
                return
            kk = self._lane_rr if k is None else int(k)
            if k is None:
                self._lane_rr = (self._lane_rr + 1) % self._lanes
            st = self._lane_streams[kk]
            cur = torch.cuda.current_stream(self._lane_dev)
            if cur.cuda_stream != 0:
                # (the legacy default stream needs no event: the lanes' streams are blocking streams and behind its work as they
                #  are - and an event recorded on it would be behind the OTHER lanes' work in flight: the lanes would take turns)
                st.wait_stream(cur)
            self._lane_pending[kk].add(cur.cuda_stream)    # (the streams that still have to join this lane)
            with torch.cuda.stream(st):
                yield kk
        return _cm()

    def join_lanes(self):
        """The current stream waits (device-side) for every lane that holds a forward its issuing stream has not joined yet; the
        forwards issued from the current stream count as joined afterwards."""
        if not self.num_lanes():
            return
        cur = torch.cuda.current_stream(self._lane_dev)
        for kk, st in enumerate(self._lane_streams):
            if self._lane_pending[kk]:
                # (every lane somebody still has to join: this stream waits for more than its own forwards at most; a join by one
                #  stream does not make the lane look joined to the others)
                cur.wait_stream(st)
                self._lane_pending[kk].discard(cur.cuda_stream)
        self.pos.handle(self._lane_dev).lanes_join(cur.cuda_stream)      # (forwards the library itself relayed from this stream)

    def _lane_of_current_stream(self, dev):
        if not self.num_lanes():
            return None
        cur = torch.cuda.current_stream(dev).cuda_stream
        for kk, st in enumerate(self._lane_streams):
            if st.cuda_stream == cur:
                return kk
        return None

    def check_status(self, device=None) -> None:
        """Synchronise and raise if a forward of the pair gave up waiting for its own tiles (the pair's flag lives in the
        pos handle; with lanes r3d_status waits for the lanes' streams too)."""
        self.pos.check_status(device)

    def checked(self, fn, device=None):
        """`fn()` (forwards through this lifter) followed by a synchronisation and the status check.  When a forward gave
        up waiting for its own tiles - the single persistent launch needs all its workgroups resident, which another
        process's kernel on the same GPU can prevent - the pair is switched to the level-by-level form for good, `fn` runs
        once more, and only a second failure raises: two processes lifting on one device both get correct outputs."""
        out = fn()
        self.join_lanes()
        dev = device if device is not None else (out.device if torch.is_tensor(out) else getattr(self.pos, "_device", None))
        try:
            self.check_status(dev)
            return out
        except _capi.Ray3DHipError:
            import warnings
            warnings.warn("ray3d_amd: a single-launch forward could not get all its workgroups resident (GPU shared?); "
                          "switching this lifter to the level-by-level form (R3D_OPT_STAGED) and repeating the call")
            self.set_staged(True)
        out = fn()
        self.join_lanes()
        self.check_status(dev)
        return out

    def _run(self, mode, x, window_stride, B, param, param_stride, cam=None, cam_stride=0,
             return_trj=False, out=None, workspace=None):
        dev = x.device
        if self.pos.training or self.trj.training:
            raise RuntimeError("ray3d_amd modules are inference-only: call .eval() first")
        if not x.is_cuda:
            raise RuntimeError("ray3d_amd runs on an AMD GPU only (got a %s tensor)" % x.device)
        hp, ht = self.pos.handle(dev), self.trj.handle(dev)
        if out is None:
            out = torch.empty((B, 1, self.pos.num_joints_in, 3), dtype=torch.float32, device=dev)
        out_trj = torch.empty((B, 1, 1, 3), dtype=torch.float32, device=dev) if return_trj else None
        if workspace is None and self.num_lanes():
            # one workspace per lane (forwards of different lanes are in flight together); a forward on a stream that is no lane's
            # is run on the next lane here, exactly as the library would relay it - so that the workspace is that lane's
            kk = self._lane_of_current_stream(dev)
            if kk is None:
                caller = torch.cuda.current_stream(dev)
                with self.lane() as k2:
                    res = self._run(mode, x, window_stride, B, param, param_stride, cam, cam_stride, return_trj, out, self._lane_ws[k2])
                    # the caller's tensors are read / written on the lane's stream: keep them alive until the lane is past this forward
                    # (a caller on a side stream that drops its input right after the call would otherwise get the block back from
                    # the caching allocator and overwrite it on its own stream while the lane still reads it).  Not record_stream on
                    # the lane's stream: the allocator would record an event there when the tensor dies - after set_lanes(0) that
                    # stream is gone.
                    done = torch.cuda.Event()
                    done.record(self._lane_streams[k2])
                    keep = self._lane_keep
                    while keep and keep[0][0].query():
                        keep.pop(0)
                    keep.append((done, x, param, cam, out))
                for t in (res if isinstance(res, tuple) else (res,)):
                    if t is not None:
                        t.record_stream(caller)        # (allocated under the lane's stream, consumed on the caller's after join_lanes)
                return res
            workspace = self._lane_ws[kk]
        ws = (workspace or self._ws).get(_capi.workspace_bytes(hp, ht, B), dev)
        inp = _capi.make_input(mode, x.data_ptr(), window_stride,
                               param.data_ptr() if param is not None else None, param_stride,
                               cam.data_ptr() if cam is not None else None, cam_stride)
        with torch.cuda.device(dev):
            _capi.forward_pair(hp, ht, inp, B, out.data_ptr(),
                               out_trj.data_ptr() if out_trj is not None else None,
                               ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
        return (out, out_trj) if return_trj else out

    def forward(self, x: torch.Tensor, param: Optional[torch.Tensor] = None, return_trj: bool = False):
        """x (B,RF,J,F) ray-encoded windows, param (B,E) -> (B,1,J,3) absolute poses."""
        self.pos._check_inputs(x, param)
        x = x.detach().to(torch.float32).contiguous()
        p = param.detach().to(x.device, torch.float32).contiguous() if self.pos.camera_embedding else None
        return self._run(_capi.R3D_INPUT_RAYS, x, self.receptive_field(), x.shape[0], p,
                         self.pos.extrinsic_dim, return_trj=return_trj)

    def forward_overlapped(self, x: torch.Tensor, param: Optional[torch.Tensor] = None, parts: int = 2):
        """forward() with the batch cut into `parts` row ranges that run concurrently on separate HIP
        streams (windows are independent).  One forward is 14 dependent launches, several of them too
        small for 256 CUs; a second, independent launch sequence fills the gaps - worth +7 % at B = 256 before the
        small launches were re-scheduled, 3 % slower than forward() since.  With the single-launch forward
        (round 3) the library orders the parts' persistent kernels one after the other (two of them must never share the
        chip), so the parts only overlap their bind / decode kernels: kept for API compatibility and for R3D_STAGED=1.  The result is ordered after the caller's stream on entry and visible to it on exit.
        Each range is scheduled for its own batch size, so sums may differ from forward() in the last
        bits (split-K tiles), never by more than the schedule-invariance tests allow."""
        self.pos._check_inputs(x, param)
        B = x.shape[0]
        parts = max(1, min(int(parts), B // 32))
        if parts == 1:
            return self.forward(x, param)
        x = x.detach().to(torch.float32).contiguous()
        p = param.detach().to(x.device, torch.float32).contiguous() if self.pos.camera_embedding else None
        dev = x.device
        while len(self._side_streams) < parts - 1:
            self._side_streams.append(torch.cuda.Stream(device=dev))
            self._side_ws.append(_Workspace())
        out = torch.empty((B, 1, self.pos.num_joints_in, 3), dtype=torch.float32, device=dev)
        cur = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(cur)
        bounds = [B * i // parts for i in range(parts + 1)]
        for i in range(parts):
            a, b = bounds[i], bounds[i + 1]
            stream = cur if i == 0 else self._side_streams[i - 1]
            if i > 0:
                stream.wait_event(ready)
            with torch.cuda.stream(stream):
                self._run(_capi.R3D_INPUT_RAYS, x[a:b], self.receptive_field(), b - a,
                          p[a:b] if p is not None else None, self.pos.extrinsic_dim, out=out[a:b],
                          workspace=None if i == 0 else self._side_ws[i - 1])
        for s in self._side_streams[: parts - 1]:
            cur.wait_stream(s)
        return out

    CLIP_CHUNK = 4096      # most windows per forward in clip mode (0.96 of the fp32-MFMA peak at this size, 0.87 at 1024)
    CLIP_ROUND = 128       # call sizes are rounded up to a multiple of this many windows (0: exact sizes)
    CLIP_BALANCED = True   # cut a long clip into near-equal calls (False: CLIP_CHUNK at a time + the rest, rounds 1-5)

    def clip_batch_sizes(self, n: int):
        """Batch sizes of the forwards that lift an n-window clip: ceil(n / CLIP_CHUNK) near-equal calls, each rounded up to a
        multiple of CLIP_ROUND (a short clip: to 1, 2, 4 ... 64 when it is that short), so that clips of any lengths share a
        handful of tile schedules (the library builds and uploads one per batch size) and no clip ends in a short,
        inefficient rest call: a 5000-window clip is 2560 + 2560, not 4096 + 1024 (the same 120 surplus windows, no short call).  The
        sum exceeds n by less than CLIP_ROUND (the last call takes what the rounded-up ones left).  CLIP_ROUND = 0 lifts exact sizes."""
        chunk, rnd = self.CLIP_CHUNK, self.CLIP_ROUND
        k = -(-n // chunk)
        sizes = []
        if k > 1:
            each = -(-n // k) if self.CLIP_BALANCED else chunk
            if rnd > 0:
                each = min(chunk, -(-each // rnd) * rnd)
            sizes = [each] * (k - 1)
        r = n - sum(sizes)
        if r > 0:
            if rnd <= 0:
                sizes.append(r)
            else:
                up = -(-r // rnd) * rnd
                # (short rests: the sizes whose schedules hold the GEMV / latency tiles of calls of a few windows - a
                #  one-window rest lifted as 32 would do 2-30 times the work)
                for small in (1, 2, 4, 8, 16, 32, 64):
                    if r <= small < up:
                        up = small
                        break
                sizes.append(up)
        return sizes

    def forward_clip(self, clip: torch.Tensor, param_row: Optional[torch.Tensor] = None):
        """clip (N + RF - 1, J, F): an edge-padded sequence; window i = frames [i, i+RF) is gathered
        in the kernels instead of materialising lib/train_val/trainer.py:47-58's copy.
        param_row (E,) is broadcast to every window (trainer.py:324).  Returns (N,1,J,3).

        The library keeps one tile schedule per batch size, and every clip has its own length: the windows are
        lifted in the batch sizes of :meth:`clip_batch_sizes` (the surplus windows slide over repeated last frames
        and their poses are cut off), so that a whole evaluation uses a handful of batch sizes."""
        rf = self.receptive_field()
        assert clip.dim() == 3 and clip.shape[1] == self.pos.num_joints_in and clip.shape[2] == self.pos.in_features
        n = clip.shape[0] - rf + 1
        if n <= 0:
            raise RuntimeError("clip shorter than the receptive field")
        clip = clip.detach().to(torch.float32).contiguous()
        p = param_row.detach().to(clip.device, torch.float32).contiguous().view(-1) \
            if self.pos.camera_embedding else None
        sizes = self.clip_batch_sizes(n)
        total = sum(sizes)
        if total == n and len(sizes) == 1:
            return self._run(_capi.R3D_INPUT_RAYS, clip, 1, n, p, 0)
        if total > n:
            clip = torch.cat([clip, clip[-1:].expand(total - n, -1, -1)], dim=0)
        out = torch.empty((total, 1, self.pos.num_joints_in, 3), dtype=torch.float32, device=clip.device)
        start = 0
        for b in sizes:
            self._run(_capi.R3D_INPUT_RAYS, clip[start:], 1, b, p, 0, out=out[start:start + b])
            start += b
        return out[:n]

    def prepare(self, batch_sizes, device=None):
        """Build and upload the tile schedules of these batch sizes now (r3d_prepare) instead of inside the first
        forward that meets them: needed before capturing a forward into a hipGraph, useful before a timed run."""
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        hp, ht = self.pos.handle(dev), self.trj.handle(dev)
        with torch.cuda.device(dev):
            for b in batch_sizes:
                _capi.prepare(hp, ht, int(b))
                self._prepared.add((dev, int(b)))

    def release_prepared(self):
        """r3d_release every batch size this lifter has prepared (nothing captured into a hipGraph may still point into them): what
        set_lanes / set_cu_limit do before they change how schedules are cut - the library refuses to drop pinned schedules."""
        for dev, b in sorted(self._prepared, key=lambda t: t[1]):
            try:
                with torch.cuda.device(dev):
                    _capi.release(self.pos.handle(dev), self.trj.handle(dev), b)
            except _capi.Ray3DHipError:
                pass
        self._prepared.clear()

    def forward_uv(self, uv: torch.Tensor, cam_rows: torch.Tensor, param: Optional[torch.Tensor] = None,
                   window_stride: Optional[int] = None):
        """Pixel keypoints in; the rays are computed inside the first layers' gathers (float64 then cast, as
        lib/camera/camera.py:423-471 + lib/train_val/trainer.py:298 do on the host) - no rays tensor exists.
        uv (B,RF,J,2) float32 - or a frame sequence (T,J,2) with `window_stride` (1: slide over an edge-padded clip);
        window i covers frames [i*stride, i*stride + RF) and is encoded with ITS camera row, also where windows
        overlap (BASELINE configs[3]: mixed intrinsics per batch);
        cam_rows (B,8) or (8,) float64 {fx,fy,cx,cy,cos(pitch),sin(pitch),0,0} (:meth:`Camera.cam_row`);
        param (B,E) or (E,) float32 [height, pitch]."""
        rf = self.receptive_field()
        uv = uv.detach().to(torch.float32).contiguous()
        if self.pos.in_features != 3:
            raise RuntimeError("forward_uv needs INPUT_DIM == 3 models (the ray encoding has three components)")
        if uv.dim() == 4:
            assert uv.shape[1] == rf and uv.shape[2] == self.pos.num_joints_in and uv.shape[3] == 2
            B, ws_ = uv.shape[0], rf
            if window_stride is not None and window_stride != rf:
                raise RuntimeError("a (B,RF,J,2) batch has window_stride == RF")
        else:
            assert uv.dim() == 3 and uv.shape[1] == self.pos.num_joints_in and uv.shape[2] == 2
            ws_ = 1 if window_stride is None else int(window_stride)
            if ws_ < 1 or uv.shape[0] < rf:
                raise RuntimeError("window_stride must be >= 1 and the sequence at least RF frames long")
            B = (uv.shape[0] - rf) // ws_ + 1
        cam = cam_rows.detach().to(uv.device, torch.float64).contiguous()
        if cam.dim() == 2 and cam.shape != (B, 8) or cam.dim() == 1 and cam.shape != (8,):
            raise RuntimeError("cam_rows must be (%d, 8) or (8,), got %s" % (B, tuple(cam.shape)))
        p = param.detach().to(uv.device, torch.float32).contiguous() if self.pos.camera_embedding else None
        pstride = 0 if (p is None or p.dim() == 1) else self.pos.extrinsic_dim
        cstride = 0 if cam.dim() == 1 else 8
        if uv.dim() == 4:
            return self._run(_capi.R3D_INPUT_UV, uv, ws_, B, p, pstride, cam, cstride)
        # a frame sequence: every clip has its own length, and the library keeps one tile schedule per batch size - the
        # windows are lifted in the batch sizes of clip_batch_sizes (as forward_clip does), the surplus ones sliding over
        # repeated last frames with the last window's camera, their poses cut off
        sizes = self.clip_batch_sizes(B)
        total = sum(sizes)
        if total == B and len(sizes) == 1:
            return self._run(_capi.R3D_INPUT_UV, uv, ws_, B, p, pstride, cam, cstride)
        if total > B:
            uv = torch.cat([uv, uv[-1:].expand((total - B) * ws_, -1, -1)], dim=0)
            if cstride:
                cam = torch.cat([cam, cam[-1:].expand(total - B, -1)], dim=0).contiguous()
            if pstride:
                p = torch.cat([p, p[-1:].expand(total - B, -1)], dim=0).contiguous()
        out = torch.empty((total, 1, self.pos.num_joints_in, 3), dtype=torch.float32, device=uv.device)
        start = 0
        for b in sizes:
            self._run(_capi.R3D_INPUT_UV, uv[start * ws_:], ws_, b, p[start:] if pstride else p, pstride,
                      cam[start:] if cstride else cam, cstride, out=out[start:start + b])
            start += b
        return out[:B]

    def profile(self, x, param):
        """One forward with per-launch HIP events; returns the launch records."""
        return self.profile_call(lambda: self.forward(x, param), x.device)

    def profile_call(self, fn, device):
        """`fn()` (one forward through this lifter, any entry point) with per-launch HIP events; the launch records."""
        h = self.pos.handle(torch.device(device))
        h.profile_enable(True)
        try:
            fn()
            return h.profile_read()
        finally:
            h.profile_enable(False)

    def last_clock_ghz(self, device) -> float:
        """The shader clock the pair's last single-launch forward ran at on `device`, in GHz (r3d_last_clock; 0.0 when it
        ran level by level).  Synchronises the current stream."""
        dev = torch.device(device)
        with torch.cuda.device(dev):
            return self.pos.handle(dev).last_clock_ghz(torch.cuda.current_stream(dev).cuda_stream)

    def precision(self, device) -> str:
        """'f32' or 'bf16x3': the arithmetic the large GEMMs of this pair run in on `device` (r3d_precision)."""
        kinds = {self.pos.handle(torch.device(device)).precision(), self.trj.handle(torch.device(device)).precision()}
        return "f32" if kinds == {"f32"} else "bf16x3"


def load_weight(model, pretrained: Dict[str, torch.Tensor]):
    """Loud replacement for lib/utils/utils.py:208-218 (which silently skips mismatching keys:
    a DataParallel checkpoint loaded into a bare model changes 0 tensors without an error).
    Accepts keys with or without the 'module.' prefix; missing/unexpected keys raise."""
    target = _unwrap(model)
    target.load_state_dict(pretrained, strict=True)
    return model


def load_checkpoint(path: str, pos_model, trj_model=None) -> dict:
    """Load a checkpoint file written by the reference's Trainer (``torch.save`` of a dict holding ``model_pos`` and,
    with a trajectory model, ``model_trj`` next to epoch / lr / optimizer state: lib/train_val/trainer.py:228-249) into
    the given modules, as main.py:190-199 does for ``--evaluate`` - strictly (see :func:`load_weight`).  ``random_state``
    in those files is a pickled NumPy generator state, so the file is unpickled in full: only open files you trust.
    Returns the rest of the dict (epoch, lr, best_performance, ...)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if "model_pos" not in ckpt:
        raise KeyError("%s has no 'model_pos' entry (keys: %s)" % (path, sorted(ckpt.keys())))
    load_weight(pos_model, ckpt["model_pos"])
    if trj_model is not None:
        if "model_trj" not in ckpt:
            raise KeyError("%s has no 'model_trj' entry but a trajectory model was given" % path)
        load_weight(trj_model, ckpt["model_trj"])
    return {k: v for k, v in ckpt.items() if k not in ("model_pos", "model_trj", "optimizer")}


_masked_streams: Dict[tuple, "torch.cuda.ExternalStream"] = {}


def _loaded_hip_runtime():
    """The HIP runtime THIS process has already loaded (the one torch drives the GPU through) as a ctypes handle: its path is taken
    from /proc/self/maps, so that no second copy of libamdhip64 (a system ROCm next to the one bundled with the wheel) gets
    initialised - a hipStream_t of another runtime instance would be an invalid handle for torch and for the library's launches."""
    import ctypes
    path = None
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
    except OSError:
        pass
    if path is None:
        raise _capi.Ray3DHipError("no HIP runtime (libamdhip64) is loaded in this process: is torch a ROCm build with a GPU visible?")
    mode = getattr(os, "RTLD_NOLOAD", 0) | getattr(os, "RTLD_NOW", 2)
    return ctypes.CDLL(path, mode=mode)


def masked_stream(cu_bits, device=None):
    """A HIP stream restricted to the CUs whose bits are set in `cu_bits` (an iterable of CU indices), as a
    torch.cuda.ExternalStream: hipExtStreamCreateWithCUMask through ctypes on the HIP runtime torch has loaded (found in
    /proc/self/maps, opened RTLD_NOLOAD).  Bit i of the mask is CU i // 8 of XCD i % 8 on MI355X's 8 x 32 CUs (the driver deals
    consecutive bits to consecutive XCDs), so `range(0, 128)` and `range(128, 256)` are two disjoint halves that both span all
    eight XCDs (and their L2s) - what R3D_OPT_CU_LIMIT needs: at least ceil(n / 8) enabled CUs in EVERY XCD, disjoint masks for
    streams used concurrently, one lifter per masked stream (include/ray3d_hip.h).  Streams are cached per (device, mask): asking
    again returns the same stream; they live until the process ends (torch does not own them)."""
    import ctypes
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    bits = tuple(sorted(set(int(b) for b in cu_bits)))
    if not bits or bits[0] < 0:
        raise ValueError("masked_stream needs at least one CU index >= 0")
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), bits)
    if key in _masked_streams:
        return _masked_streams[key]
    words = (bits[-1] // 32) + 1
    mask = (ctypes.c_uint32 * words)()
    for b in bits:
        mask[b // 32] |= 1 << (b % 32)
    hip = _loaded_hip_runtime()
    stream = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), ctypes.c_uint32(words), mask)
    if rc != 0 or not stream.value:
        raise _capi.Ray3DHipError("hipExtStreamCreateWithCUMask failed (%d)" % rc)
    _masked_streams[key] = torch.cuda.ExternalStream(stream.value, device=dev)
    return _masked_streams[key]
