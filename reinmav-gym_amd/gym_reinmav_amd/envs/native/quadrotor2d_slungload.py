"""``quadrotor2d-slungload-v0`` - drop-in for ``Quadrotor2DSlungload`` (quadrotor2d_slungload.py:41-188)."""
import numpy as np

from .base import NativeQuadrotorEnv


class Quadrotor2DSlungload(NativeQuadrotorEnv):
    _kind = "quad2d_sl"
    _action_box = (-10.0, 10.0, np.float32)  # quadrotor2d_slungload.py:68
