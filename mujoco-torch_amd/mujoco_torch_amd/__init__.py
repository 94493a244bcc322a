"""MI355X-native batched physics stepper behind the ``mujoco_torch.step`` API.

Drop-in for the hot path of vmoens/mujoco-torch (``mujoco_torch/__init__.py:41-136``):
``step``, ``forward``, ``device_put``, ``make_data`` and the ``Model`` / ``Data`` / ``Contact`` /
``Option`` schema.  Use as ``import mujoco_torch_amd as mujoco_torch``.
"""

from . import mjcf  # noqa: F401
from ._enums import (  # noqa: F401
    BiasType,
    CamLightType,
    ConeType,
    DisableBit,
    DynType,
    EnableBit,
    EqType,
    GainType,
    GeomType,
    IntegratorType,
    JacobianType,
    JointType,
    SolverType,
    TrnType,
)
from .container import MjTensorClass, UnbatchedTensor  # noqa: F401
from .device import device_get_into, device_put  # noqa: F401
from .forward import forward, reset_where, step  # noqa: F401
from .io import make_data  # noqa: F401
from .types import Contact, Data, Model, Option, Statistic  # noqa: F401

__version__ = "0.2.0"


def test_data_path(name: str = "") -> str:
    """Path of a bundled model file (``mujoco_torch_amd/test_data/``: the reference's ``mujoco_torch/test_data`` models plus
    the scenes of BASELINE.json's configs), e.g. ``test_data_path("humanoid.xml")``."""
    import os

    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_data", name)
