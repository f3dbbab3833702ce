"""Real-data front end: the pose archives the reference evaluates on -> :class:`evaluate.Clip` lists.

Counterpart of the evaluation-side work of ``lib/dataset/__init__.py`` (``Data``: :49-203 load, ray-encode and
sanity-check; :279-330 ``fetch_via_action``) and of the camera construction in the dataset classes
(``lib/dataset/h36m_dataset.py:349-386``, ``lib/dataset/mpii_3dhp_dataset.py:309-341``), reduced to what the lifting
path consumes:

* the 3D archive: ``np.load(path, allow_pickle=True)['positions_3d'].item()`` = ``{subject: {action: (N, J3, 3)}}``
  world coordinates in metres (:386-395 / :345-353);
* the 2D archive: ``['positions_2d'].item()`` = ``{subject: {action: [per-camera (N2, J, 2) pixel keypoints]}}`` - or,
  in the 3DHP ground-truth file, per-camera dicts with ``'positions_2d'`` - plus ``['metadata'].item()`` with
  ``keypoints_symmetry`` (``lib/dataset/__init__.py:112-122``);
* per camera i of a subject: ground truth = ``camera.world2normalized(positions)`` (:94-108), model input =
  ``camera.get_cam_ray_given_uv(keypoints)`` (:191-203), the 2D sequence cut to the mocap length (:205-231).

The calibration tables themselves (numbers in the reference's dataset modules) are not part of this package: hand
them to :func:`cameras_from_tables` in the reference's own dict layout - e.g. ``h36m_cameras_extrinsic_params`` and
``h36m_cameras_intrinsic_params`` imported from a Ray3D checkout, or a JSON dump of them.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

from .camera import Camera
from .evaluate import Clip

# Index sets of the dataset conventions (joints of the 32-joint H36M mocap skeleton that are kept).
H36M_32_TO_17 = (0, 1, 2, 3, 6, 7, 8, 12, 13, 14, 15, 17, 18, 19, 25, 26, 27)   # h36m_dataset.py:403-404 complement
KEEP_UNIVERSAL_14_OF_17 = (0, 1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15, 16)      # h36m_dataset.py:430, mpii_3dhp:357
SYMMETRY_17 = ((4, 5, 6, 11, 12, 13), (1, 2, 3, 14, 15, 16))                    # joints_left, joints_right
SYMMETRY_14 = ((4, 5, 6, 8, 9, 10), (1, 2, 3, 11, 12, 13))                      # h36m_dataset.py:432
SYMMETRY_HUMANEVA_15 = ((2, 3, 4, 8, 9, 10), (5, 6, 7, 11, 12, 13))             # humaneva_dataset.py:7-9 (joints_left, joints_right)
HUMANEVA_15_TO_UNIVERSAL_14 = (0, 11, 12, 13, 8, 9, 10, 14, 2, 3, 4, 5, 6, 7)   # humaneva_dataset.py:126,150: order AND selection
# Subjects of the camera-augmented H36M set: the seven originals and their copies scaled by 0.6 ... 1.1
# (lib/dataset/h36m_aug_dataset.py:26-34); every one of them is seen by every camera of the JSON list.
H36M_AUG_SCALES = ("0.6", "0.7", "0.8", "0.9", "1.1")
H36M_AUG_SUBJECTS = tuple(["S1", "S5", "S6", "S7", "S8", "S9", "S11"] +
                          ["%s_%s" % (s, k) for k in H36M_AUG_SCALES for s in ("S1", "S5", "S6", "S7", "S8", "S9", "S11")])


def cameras_from_tables(extrinsics: Mapping[str, Sequence[Mapping]], intrinsics: Optional[Sequence[Mapping]] = None,
                        translation_divisor: float = 1, undistort: bool = False) -> Dict[str, List[Camera]]:
    """``{subject: [Camera per view]}`` from calibration tables in the reference's layout.

    ``extrinsics[subject][i]`` holds ``'R'`` (3x3 world->camera) and ``'translation'`` (3,), optionally with the
    intrinsic keys merged in (the 3DHP table); ``intrinsics[i]`` holds ``'focal_length'``, ``'center'`` and, for
    ``undistort=True``, ``'radial_distortion'`` (k1, k2, k3) and ``'tangential_distortion'`` (p1, p2).  Every number
    goes through float32 first, as the dataset classes do (``np.array(v, dtype='float32')``, h36m_dataset.py:355-359);
    H36M translations are in millimetres there: pass ``translation_divisor=1000`` (:361-363, divided in float32).
    Entries without a translation are skipped (:369-370)."""
    f32 = lambda v: np.array(v, dtype="float32")
    out: Dict[str, List[Camera]] = {}
    for subject, views in extrinsics.items():
        cams = []
        for i, ext in enumerate(views):
            cam = dict(ext)
            if intrinsics is not None:
                cam.update(intrinsics[i])
            if "translation" not in cam:
                continue
            fl, ce = f32(cam["focal_length"]), f32(cam["center"])
            K = np.eye(3, dtype=np.float64)
            K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fl[0], fl[1], ce[0], ce[1]
            t = np.array(f32(cam["translation"]) / translation_divisor, dtype=np.float64)
            dist = None
            if undistort:
                rad, tan = f32(cam["radial_distortion"]), f32(cam["tangential_distortion"])
                dist = np.concatenate((rad[:2], tan, rad[2:])).astype(np.float32)      # (k1, k2, p1, p2, k3), :379-381
            cams.append(Camera(K, f32(cam["R"]), t, dist_coeff=dist, undistort=undistort,
                               name="%s/%d" % (subject, i)))
        out[subject] = cams
    return out


def cameras_from_json(camera_meta, subjects: Sequence[str] = H36M_AUG_SUBJECTS) -> Tuple[Dict[str, List[Camera]], List[str]]:
    """The camera-augmented H36M front end (lib/dataset/h36m_aug_dataset.py:21-60): ONE list of cameras - the JSON file
    ``data/aggregate_camera.py`` writes from the per-camera files of ``data/camera_augmentation.py:720-731``, each entry
    holding ``id, center, focal_length, radial_distortion, tangential_distortion, res_w, res_h, azimuth, R, translation`` -
    shared by every subject of the set (the originals and the scaled copies ``S1_0.9`` ...).  ``camera_meta`` is that
    list or the path of the JSON file.  Unlike the plain H36M tables nothing is rounded to float32 and the translation
    is already in metres (:44-53); the cameras are built with ``undistort=False`` (:57-59).
    Returns ``({subject: [Camera per JSON entry]}, camera ids)`` - the ids are ``camera_dist`` of
    ``CAMERA_WISE_PERFORMANCE`` (:36-41), in file order."""
    if isinstance(camera_meta, (str, bytes)):
        import json
        with open(camera_meta, "r") as fh:
            camera_meta = json.load(fh)
    per_cam = []
    ids = []
    for cam in camera_meta:
        K = np.eye(3, dtype=np.float64)
        K[0, 0], K[1, 1] = cam["focal_length"][0], cam["focal_length"][1]
        K[0, 2], K[1, 2] = cam["center"][0], cam["center"][1]
        R = np.array(cam["R"], dtype=np.float64).reshape(3, 3)
        t = np.array(cam["translation"], dtype=np.float64).reshape(3, 1)
        dist = np.array(list(cam["radial_distortion"][:2]) + list(cam["tangential_distortion"]) +
                        list(cam["radial_distortion"][2:]), dtype=np.float64).reshape(5)
        per_cam.append((K, R, t, dist))
        ids.append(str(cam["id"]))
    out = {s: [Camera(K, R, t, dist_coeff=d, undistort=False, name="%s/%s" % (s, cid))
               for (K, R, t, d), cid in zip(per_cam, ids)] for s in subjects}
    return out, ids


def cameras_humaneva(extrinsics: Mapping[str, Sequence[Mapping]], intrinsics: Sequence[Mapping],
                     undistort: bool = True) -> Dict[str, List[Camera]]:
    """HumanEva-I's layout (lib/dataset/humaneva_dataset.py:70-106): per-subject extrinsic tables ('S1', 'S2', 'S3'),
    one intrinsic table per camera index, numbers through float32 and translations from millimetres (:76-82), and every
    subject's cameras registered under BOTH archive prefixes, ``'Train/S1'`` and ``'Validate/S1'`` (:100-104).  The
    reference builds these cameras with CameraInfoPacket's default ``undistort=True``; that path is the OpenCV
    restatement of :meth:`Camera.undistort_points` (parity unpinned)."""
    base = cameras_from_tables(extrinsics, intrinsics, translation_divisor=1000, undistort=undistort)
    out: Dict[str, List[Camera]] = {}
    for subject, cams in base.items():
        for prefix in ("Train/", "Validate/"):
            out[prefix + subject] = cams
    return out


@dataclass
class PoseData:
    """What ``Data`` + ``fetch_via_action`` hand to the evaluation loop, as clips."""
    clips: List[Clip]
    subjects: List[str]
    kps_left: List[int]
    kps_right: List[int]
    joints_left: List[int]
    joints_right: List[int]
    actions: Dict[str, List[int]] = field(default_factory=dict)    # action key ('Walking' of 'Walking 1') -> clip ids
    camera_index: List[int] = field(default_factory=list)          # per clip: index of its camera in the subject's list


def _per_camera_keypoints(entry) -> np.ndarray:
    # plain array, or the 3DHP ground-truth file's {'positions_2d': ..., 'file_name': ...} (mpii_3dhp_dataset.py:398)
    return np.asarray(entry["positions_2d"] if isinstance(entry, dict) else entry)


def load_pose_data(path_3d: str, path_2d: str, cameras: Mapping[str, Sequence[Camera]], subjects: Sequence[str],
                   joints_3d: Optional[Sequence[int]] = None, joints_2d: Optional[Sequence[int]] = None,
                   action_filter: Optional[Sequence[str]] = None, downsample: int = 1,
                   joints_symmetry: Optional[Tuple[Sequence[int], Sequence[int]]] = None) -> PoseData:
    """Read the two archives and build one :class:`Clip` per (subject, action, camera) - the sequences
    ``Trainer.evaluate`` walks (lib/train_val/trainer.py:407-460), grouped by ``action.split(' ')[0]`` (:417).

    ``joints_3d`` / ``joints_2d`` select joints from the archives (``H36M_32_TO_17`` for the 32-joint H36M mocap file,
    ``KEEP_UNIVERSAL_14_OF_17`` for the cross-dataset 14-joint layout); ``action_filter`` is main.py's ``ACTIONS`` list used the way
    ``Trainer.evaluate`` uses it (:412-417): its entries are EXACT action keys looked up for every test subject
    ('Walking' selects the sequence called 'Walking', not 'Walking 1' or 'WalkingDog'; a key a subject lacks raises
    ``KeyError`` as the reference's ``fetch_via_action`` would); ``downsample`` is ``DOWNSAMPLE`` as
    ``fetch_via_action`` applies it (lib/dataset/__init__.py:342-348);
    ``joints_symmetry`` = the skeleton's (joints_left, joints_right) when it is neither the 17- nor the 14-joint one.
    Raises on what ``sanity_check`` (:205-231) asserts: missing subject/action, fewer 2D frames than mocap frames,
    camera-count mismatch."""
    a3 = np.load(path_3d, allow_pickle=True)["positions_3d"].item()
    z2 = np.load(path_2d, allow_pickle=True)
    a2, meta = z2["positions_2d"].item(), z2["metadata"].item()
    sym = meta["keypoints_symmetry"]
    kps_left, kps_right = [int(v) for v in sym[0]], [int(v) for v in sym[1]]
    if joints_2d is not None:
        # the universal layout renumbers the kept joints (h36m_dataset.py:430-432)
        pos = {int(j): k for k, j in enumerate(joints_2d)}
        kps_left = [pos[j] for j in kps_left if j in pos]
        kps_right = [pos[j] for j in kps_right if j in pos]
    clips: List[Clip] = []
    groups: Dict[str, List[int]] = {}
    cam_of_clip: List[int] = []
    n_joints = None
    for subject in subjects:
        if subject not in a3:
            raise KeyError("subject %r is missing from the 3D archive %s" % (subject, path_3d))
        if subject not in a2:
            raise KeyError("Subject %s is missing from the 2D detections dataset" % subject)
        if subject not in cameras:
            raise KeyError("no cameras given for subject %r" % subject)
        if action_filter is not None:
            for a in action_filter:
                if a not in a3[subject]:
                    raise KeyError("action %r (ACTIONS) is not a sequence of subject %r: the filter's entries are exact "
                                   "action keys (lib/train_val/trainer.py:412-417)" % (a, subject))
            wanted = [(a, a3[subject][a]) for a in action_filter]
        else:
            wanted = list(a3[subject].items())
        for action, pos3d in wanted:
            if action not in a2[subject]:
                raise KeyError("Action %s of subject %s is missing from the 2D detections dataset" % (action, subject))
            views = a2[subject][action]
            cams = cameras[subject]
            if len(views) != len(cams):
                raise ValueError("Camera count mismatch for %s / %s: %d keypoint sequences, %d cameras"
                                 % (subject, action, len(views), len(cams)))
            world = np.asarray(pos3d)
            if joints_3d is not None:
                world = world[:, list(joints_3d)]
            for ci, (cam, entry) in enumerate(zip(cams, views)):
                kps = _per_camera_keypoints(entry)[..., :2]
                if joints_2d is not None:
                    kps = kps[:, list(joints_2d)]
                if kps.shape[0] < world.shape[0]:
                    raise ValueError("%s / %s camera %d: %d keypoint frames for %d mocap frames"
                                     % (subject, action, ci, kps.shape[0], world.shape[0]))
                kps = kps[:world.shape[0]]                       # some videos carry extra frames (:222-228)
                if kps.shape[1] != world.shape[1]:
                    raise ValueError("%s / %s: %d keypoints vs %d joints" % (subject, action, kps.shape[1], world.shape[1]))
                n_joints = world.shape[1]
                gt = cam.world2normalized(world)[::downsample]
                rays = cam.rays_from_uv(kps)[::downsample]
                key = action.split(" ")[0]
                cid = len(clips)
                clips.append(Clip(cam, rays.astype(np.float32), gt.astype(np.float32), key, cid))
                groups.setdefault(key, []).append(cid)
                cam_of_clip.append(ci)
    if joints_symmetry is not None:
        jl, jr = joints_symmetry
    elif n_joints in (14, 17, None):
        jl, jr = SYMMETRY_14 if n_joints == 14 else SYMMETRY_17
    else:
        raise ValueError("give joints_symmetry=(left, right) for a %d-joint skeleton (HumanEva's 15 joints: "
                         "dataset.SYMMETRY_HUMANEVA_15)" % n_joints)
    return PoseData(clips, list(subjects), kps_left, kps_right, list(jl), list(jr), groups, cam_of_clip)
