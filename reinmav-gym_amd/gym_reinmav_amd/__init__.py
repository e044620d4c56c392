"""gym_reinmav_amd - MI355X-native batched drop-in for reinmav-gym's native quadrotor envs.

Single envs (gym.Env-shaped, batch = 1):   ``make('quadrotor3d-v0')`` / ``envs.native.Quadrotor3D``
Batched (baselines VecEnv-shaped):         ``QuadrotorVecEnv('quadrotor3d-v0', num_envs=65536)``
Low level (rollouts, state access, stats): ``BatchedQuadrotor('quad3d', 65536)``
All arithmetic runs in the HIP kernels of ``librmav.so``; there is no CPU fallback.
"""
from . import _abi
from ._abi import RmavError
from .core import BatchedQuadrotor
from .registration import ENTRY_POINTS, make, register_envs
from .vec_env import ENV_IDS, QuadrotorVecEnv

register_envs()

__all__ = ["BatchedQuadrotor", "QuadrotorVecEnv", "RmavError", "make", "register_envs", "ENTRY_POINTS", "ENV_IDS",
           "_abi"]
__version__ = "0.1.0"
