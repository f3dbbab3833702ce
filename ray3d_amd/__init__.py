"""ray3d_amd - MI355X-native implementation of Ray3D's 2D->3D lifting forward pass.

Public surface (drop-in for the reference's lifting path):

    from ray3d_amd import Model                 # lib/model/__init__.py:5   factory
    from ray3d_amd import RIEModel, RIETrajectoryModel, Ray3DLifter
    from ray3d_amd import Camera                # lib/camera/camera.py:208  per-camera constants
    from ray3d_amd import evaluate              # lib/train_val/trainer.py:283 evaluation loop
    from ray3d_amd import dataset               # lib/dataset/__init__.py:49  pose archives -> clips

All arithmetic of the networks runs in libray3d_hip.so (hand-written gfx950 kernels behind the C
ABI of include/ray3d_hip.h).  No CPU fallback exists.
"""
from .spec import LiftConfig, config_from_dicts, default_model_config   # noqa: F401
from .modules import (Model, RIEModel, RIETrajectoryModel, Ray3DLifter, load_checkpoint, load_weight, masked_stream)   # noqa: F401
from .camera import Camera, augment_camera, camera_grid, synthetic_camera   # noqa: F401
from . import dataset, evaluate, metrics, synth   # noqa: F401

__version__ = "0.1.0"
