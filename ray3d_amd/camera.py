"""Per-camera constants of the ray-encoding front end (host side, float64).

Counterpart of the parts of ``CameraInfoPacket`` that sit on the lifting path
(lib/camera/camera.py:210-277 constructor, :308-316 pitch, :325-345 normalised frame,
:390-410 world<->normalised, :423-471 uv -> ray).  Everything per-camera is a handful of float64
numbers computed once on the host; everything per-keypoint happens on the GPU
(`Ray3DLifter.forward_uv`: the first-level gather of r3d_gemm_f32 encodes the rays it stages), fed by :meth:`Camera.cam_row`.

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
        if undistort and self.dist_coeff is None:
            raise ValueError("undistort=True needs dist_coeff (k1, k2, p1, p2, k3)")
        # undistort=True (the H36M default, lib/dataset/h36m_dataset.py:383-385): pixel keypoints go through
        # undistort_points() before the ray encoding, as lib/camera/camera.py:423-441 does.  PARITY UNPINNED: the
        # reference calls cv2.undistortPoints (opencv-python 4.4.0.42), absent here and pinned by no reference
        # test; undistort_points restates OpenCV's documented algorithm (see its docstring).
        self.undistort = bool(undistort)
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
    def undistort_points(self, uv: np.ndarray) -> np.ndarray:
        """Pixel keypoints (..., 2) with the lens distortion removed, float64: the counterpart of
        CameraInfoPacket.undistort_point (lib/camera/camera.py:412-421), i.e.
        ``cv2.undistortPoints(points2d, K, dist_coeff, P=K)``.  PARITY UNPINNED - OpenCV is a third-party
        dependency absent from the reference tree and from this image; this is OpenCV's documented method for
        the 5-coefficient Brown-Conrady model (k1, k2, p1, p2, k3): five fixed-point iterations of
        x <- (x0 - tangential(x)) / radial(x) on normalised coordinates, then re-projection with K.  Checked only
        by the distort(undistort(p)) round trip and the principal-point fixed point (tests/test_host.py)."""
        if self.dist_coeff is None:
            raise ValueError("this camera has no distortion coefficients")
        k1, k2, p1, p2, k3 = (float(v) for v in self.dist_coeff)
        uv = np.asarray(uv, dtype=np.float64)
        x0 = (uv[..., 0] - self.cx) / self.fx
        y0 = (uv[..., 1] - self.cy) / self.fy
        x, y = x0.copy(), y0.copy()
        for _ in range(5):
            r2 = x * x + y * y
            icd = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
            dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
            dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
            x = (x0 - dx) * icd
            y = (y0 - dy) * icd
        return np.stack([x * self.fx + self.cx, y * self.fy + self.cy], axis=-1)

    def distort_points(self, uv: np.ndarray) -> np.ndarray:
        """Forward Brown-Conrady model on ideal pixels (data/camera_augmentation.py:502-542), for round trips."""
        k1, k2, p1, p2, k3 = (float(v) for v in self.dist_coeff)
        uv = np.asarray(uv, dtype=np.float64)
        x = (uv[..., 0] - self.cx) / self.fx
        y = (uv[..., 1] - self.cy) / self.fy
        r2 = x * x + y * y
        rad = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
        xd = x * rad + (2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x))
        yd = y * rad + (p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y)
        return np.stack([xd * self.fx + self.cx, yd * self.fy + self.cy], axis=-1)

    def pixels_for_encoding(self, uv: np.ndarray) -> np.ndarray:
        """What encode_uv_with_intrinsic (camera.py:423-441) subtracts the principal point from: the keypoints,
        undistorted first when the camera was built with undistort=True.  Feed this to Ray3DLifter.forward_uv."""
        return self.undistort_points(uv) if self.undistort else np.asarray(uv, dtype=np.float64)

    def rays_from_uv(self, uv: np.ndarray) -> np.ndarray:
        uv = self.pixels_for_encoding(uv)
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


def _rodrigues(axis, radian: float) -> np.ndarray:
    """Rotation by `radian` about `axis` - what data/camera_augmentation.py:433-441 computes as
    expm(cross(I, axis / |axis| * radian))."""
    a = np.asarray(axis, dtype=np.float64).reshape(3)
    a = a / np.linalg.norm(a)
    Kx = np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])
    return np.eye(3) + math.sin(radian) * Kx + (1.0 - math.cos(radian)) * (Kx @ Kx)


def _rotate_camera(R, T, center, axis, radian):
    """data/camera_augmentation.py:443-466: rotate the camera (pose and position) about `axis` through `center`."""
    Rc2w = R.T
    Tc2w = -Rc2w @ T
    M = _rodrigues(axis, radian)
    new_Rc2w = M @ Rc2w
    new_Tc2w = M @ (Tc2w - center) + center
    new_Rw2c = new_Rc2w.T
    return new_Rw2c, -new_Rw2c @ new_Tc2w


def augment_camera(R, t, yaw_deg: float, dist_ratio: float, pitch_deg: float, center=(0.0, 0.0, 1.8)):
    """One camera of the reference's synthetic camera sets (data/camera_augmentation.py:696-718) from a base camera
    (R world->camera, t in metres): the translation scaled about the centre point ((T - c) * ratio + c, applied to
    Tw2c exactly as the script does, :416-424), a yaw about the world z axis through the centre, then a pitch about
    the horizontal axis perpendicular to the camera's position vector.  Returns (R, t (3,1)) of the new camera."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    T = np.asarray(t, dtype=np.float64).reshape(3, 1)
    c = np.asarray(center, dtype=np.float64).reshape(3, 1)
    T1 = (T - c) * dist_ratio + c
    R2, T2 = _rotate_camera(R, T1, c, np.array([0.0, 0.0, 1.0]), yaw_deg / 180.0 * np.pi)
    pos = -R2.T @ T2
    axis = np.array([-pos[1, 0], pos[0, 0], 0.0])
    return _rotate_camera(R2, T2, c, axis, pitch_deg / 180.0 * np.pi)


# The 'Train' set of data/camera_augmentation.py:637-642: yaw x distance ratio x pitch (degrees), 342 cameras
AUGMENTATION_TRAIN_GRID = ((60, 180, 300), (2.0, 2.2, 2.4, 2.6, 2.8, 3.0),
                           tuple(range(-26, 11, 2)))


def camera_grid(K, R, t, grid=AUGMENTATION_TRAIN_GRID, center=(0.0, 0.0, 1.8)):
    """All cameras of a (yaws, distance ratios, pitches) grid around a base camera, in the script's loop order
    (yaw outermost, pitch innermost; :680-688)."""
    cams = []
    for yaw in grid[0]:
        for dist in grid[1]:
            for pitch in grid[2]:
                Rn, tn = augment_camera(R, t, yaw, dist, pitch, center)
                cams.append(Camera(K, Rn, tn, name="yaw%g_d%g_p%g" % (yaw, dist, pitch)))
    return cams
