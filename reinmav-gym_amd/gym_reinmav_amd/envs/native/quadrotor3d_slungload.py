"""``quadrotor3d-slungload-v0`` - drop-in for ``Quadrotor3DSlungload`` (quadrotor3d_slungload.py:42-231)."""
import numpy as np

from .base import NativeQuadrotorEnv


class Quadrotor3DSlungload(NativeQuadrotorEnv):
    _kind = "quad3d_sl"
    _action_box = (-10.0, 10.0, np.float32)  # quadrotor3d_slungload.py:75
