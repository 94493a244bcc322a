"""Environment zoo on the native stepper (reference ``mujoco_torch/zoo/__init__.py``): ``ENVS["halfcheetah"](num_envs=64)``."""

from . import envs as _envs  # noqa: F401  (fills the registry)
from .base import ENVS, MujocoTorchEnv, register_env

globals().update({cls.__name__: cls for cls in ENVS.values()})
__all__ = ["ENVS", "MujocoTorchEnv", "register_env", *[cls.__name__ for cls in ENVS.values()]]
