"""``quadrotor2d-v0`` - drop-in for ``Quadrotor2D`` (quadrotor2d.py:40-142).

The reference file has a syntax error in its termination test (quadrotor2d.py:95-98).  Reading "B"
(default: add the one missing line-continuation; done = |p|>3 or |v|>2) and reading "A" (drop the
dangling clauses; done = |p|>3 or |v|>10) are both available: ``Quadrotor2D(reading='A')``."""
import numpy as np

from .base import NativeQuadrotorEnv


class Quadrotor2D(NativeQuadrotorEnv):
    _kind = "quad2d"
    _action_box = (-10.0, 10.0, np.float32)  # quadrotor2d.py:62

    def __init__(self, device: int = 0, seed=None, reading: str = "B"):
        self._reading_2d = reading
        super().__init__(device=device, seed=seed)
