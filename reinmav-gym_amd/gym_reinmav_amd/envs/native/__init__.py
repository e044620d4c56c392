"""Same class names as ``gym_reinmav.envs.native`` (ReinmavEnv is out of scope)."""
from .quadrotor2d import Quadrotor2D
from .quadrotor2d_slungload import Quadrotor2DSlungload
from .quadrotor3d import Quadrotor3D
from .quadrotor3d_slungload import Quadrotor3DSlungload

__all__ = ["Quadrotor2D", "Quadrotor2DSlungload", "Quadrotor3D", "Quadrotor3DSlungload"]
