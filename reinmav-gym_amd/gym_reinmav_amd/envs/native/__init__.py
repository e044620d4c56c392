"""Same class names as ``gym_reinmav.envs.native`` (gym_reinmav/envs/native/__init__.py:1-5)."""
from .quadrotor2d import Quadrotor2D
from .quadrotor2d_slungload import Quadrotor2DSlungload
from .quadrotor3d import Quadrotor3D
from .quadrotor3d_slungload import Quadrotor3DSlungload
from .reinmav_env import ReinmavEnv

__all__ = ["ReinmavEnv", "Quadrotor2D", "Quadrotor2DSlungload", "Quadrotor3D", "Quadrotor3DSlungload"]
