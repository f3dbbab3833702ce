"""Per-camera constants of the ray-encoding front end (host side, float64).

Counterpart of the parts of ``CameraInfoPacket`` that sit on the lifting path
(lib/camera/camera.py:210-277 constructor, :308-316 pitch, :325-345 normalised frame,
:390-410 world<->normalised, :423-471 uv -> ray).  Everything per-camera is a handful of float64
numbers computed once on the host; everything per-keypoint happens on the GPU
(`Ray3DLifter.forward_uv`, kernel r3d_encode_f32), fed by :meth:`Camera.cam_row`.

"Normalised" follows the reference's meaning: the camera frame rotated about its
x axis by the camera pitch and shifted by the camera height - NOT unit-length rays.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np


class Camera:
    def __init__(self, K, R, t, dist_coeff=None, undistort: bool = False, name: str = ""):
        """K (3,3) intrinsics, R (3,3) world->camera rotation, t (3,) or (3,1) translation in the
        camera frame (metres), exactly the arguments of CameraInfoPacket(K=, R=, t=)."""
        self.name = name
        self.K = np.asarray(K, dtype=np.float64).reshape(3, 3)
        self.Rw2c = np.asarray(R, dtype=np.float64).reshape(3, 3)
        self.Tw2c = np.asarray(t, dtype=np.float64).reshape(3, 1)
        self.dist_coeff = None if dist_coeff is None else np.asarray(dist_coeff, dtype=np.float64).reshape(5)
        if undistort:
            # cv2.undistortPoints (opencv-python 4.4, camera.py:420) is not available offline and no
            # reference fixture pins it; refuse rather than ship an unpinned approximation.
            raise NotImplementedError("undistort=True needs OpenCV's undistortPoints; encode H36M "
                                      "keypoints with undistort=False or undistort them upstream")
        self.undistort = False
        self.fx, self.fy = self.K[0, 0], self.K[1, 1]
        self.cx, self.cy = self.K[0, 2], self.K[1, 2]

        rc2w = self.Rw2c.T
        self.position_world = -(rc2w @ self.Tw2c)                 # camera centre in world coords
        self.height = float(self.position_world[2, 0])            # param[0]
        axis = rc2w[:, 2]                                         # optical axis in world coords
        self.pitch = math.acos(axis[2] / math.sqrt(float(axis @ axis))) - math.pi / 2   # param[1]
        self.cos_p, self.sin_p = math.cos(self.pitch), math.sin(self.pitch)

        self.Rc2n = np.array([[1.0, 0.0, 0.0],
                              [0.0, self.cos_p, self.sin_p],
                              [0.0, -self.sin_p, self.cos_p]], dtype=np.float64)
        self.Tc2n = np.array([[0.0], [-self.height], [0.0]], dtype=np.float64)
        self.Rw2n = self.Rc2n @ self.Rw2c
        self.Tw2n = self.Rc2n @ self.Tw2c + self.Tc2n
        self.Rn2w = rc2w @ self.Rc2n.T
        self.Tn2w = -self.Rn2w @ self.Tc2n - rc2w @ self.Tw2c

    # -- what the networks consume
    def param(self) -> np.ndarray:
        """[camera height (m), pitch (rad)] float32 - lib/train_val/trainer.py:297."""
        return np.array([self.height, self.pitch], dtype=np.float32)

    def cam_row(self) -> np.ndarray:
        """float64 row for r3d_input.cam_dev: {fx, fy, cx, cy, cos(pitch), sin(pitch), 0, 0}."""
        return np.array([self.fx, self.fy, self.cx, self.cy, self.cos_p, self.sin_p, 0.0, 0.0],
                        dtype=np.float64)

    # -- host-side equivalents (dataset-load time in the reference, lib/dataset/__init__.py:191-203)
    def rays_from_uv(self, uv: np.ndarray) -> np.ndarray:
        uv = np.asarray(uv, dtype=np.float64)
        x = (uv[..., 0] - self.cx) / self.fx
        y = (uv[..., 1] - self.cy) / self.fy
        return np.stack([x, self.cos_p * y + self.sin_p, -self.sin_p * y + self.cos_p], axis=-1)

    def uv_from_rays(self, rays: np.ndarray) -> np.ndarray:
        pc = np.asarray(rays, dtype=np.float64) @ self.Rc2n          # rays @ Rn2c^T, Rn2c = Rc2n^T
        return np.stack([pc[..., 0] * self.fx + self.cx, pc[..., 1] * self.fy + self.cy], axis=-1)

    def world2normalized(self, pts):
        return np.asarray(pts, dtype=np.float64) @ self.Rw2n.T + self.Tw2n.T

    def normalized2world(self, pts):
        return np.asarray(pts, dtype=np.float64) @ self.Rn2w.T + self.Tn2w.T

    def project(self, pts_world):
        """Pinhole projection of (...,3) world points to pixels (camera.py:485-504, no distortion)."""
        pc = np.asarray(pts_world, dtype=np.float64) @ self.Rw2c.T + self.Tw2c.T
        return np.stack([pc[..., 0] / pc[..., 2] * self.fx + self.cx,
                         pc[..., 1] / pc[..., 2] * self.fy + self.cy], axis=-1)


def synthetic_camera(yaw_deg: float, distance: float, pitch_deg: float, height: Optional[float] = None,
                     focal: float = 1145.0, center=(512.0, 512.0), name: str = "") -> Camera:
    """A camera on a circle around the origin looking at a point 1 m above the ground - the shape
    of the reference's synthetic camera sweep (data/camera_augmentation.py:637-664: yaw x distance
    x pitch grids).  World frame: z up, ground at z=0."""
    yaw, pit = math.radians(yaw_deg), math.radians(pitch_deg)
    h = height if height is not None else 1.0 + distance * math.tan(-pit) if pit < 0 else 1.0 + 0.5
    pos = np.array([distance * math.cos(yaw), distance * math.sin(yaw), h])
    fwd = np.array([-math.cos(yaw) * math.cos(pit), -math.sin(yaw) * math.cos(pit), math.sin(pit)])
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=0)           # rows = camera axes in world coords
    t = -R @ pos
    K = np.array([[focal, 0.0, center[0]], [0.0, focal, center[1]], [0.0, 0.0, 1.0]])
    return Camera(K, R, t, name=name)
