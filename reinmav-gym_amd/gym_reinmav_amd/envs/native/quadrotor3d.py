"""``quadrotor3d-v0`` - drop-in for the reference's ``Quadrotor3D`` (quadrotor3d.py:42-185)."""
import numpy as np

from .base import NativeQuadrotorEnv


class Quadrotor3D(NativeQuadrotorEnv):
    _kind = "quad3d"
    _action_box = (0.0, 10.0, np.float64)  # quadrotor3d.py:70 Box(0, 10, dtype=np.float)
