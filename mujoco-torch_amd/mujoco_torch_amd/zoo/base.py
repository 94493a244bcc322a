"""Batched environment on top of the native stepper -- the reference's TorchRL caller of the hot path.

Mirrors ``MujocoTorchEnv`` of reference ``zoo/base.py`` (constructor arguments :62-76, subclass hooks :164-205, XML patch
:235-264, ``_make_batch`` :266-273, ``_reset`` :275-305, ``_step`` :307-343): same class attributes, same hook names, same
TensorDict keys, same order of operations inside a step (ctrl -> frame_skip physics steps -> reward / termination from the
terminal state -> observation -> fused auto-reset).

What is different, by design:
  * the physics step is ``frame_skip`` native launches over two resident ping-pong ``Data`` buffers instead of
    ``torch.vmap(step)`` (optionally compiled);
  * a masked reset (``self._dx[mask] = self._make_batch(n)``) is ONE native launch (``reset_where``, csrc/mjh_reset.h) with
    no host sync: no ``mask.any()``, no ``int(mask.sum())``, no gather of n fresh environments.  The reset states are drawn
    for every environment (``dx0 + U(-noise, noise)``, two elementwise launches) and only the masked rows are consumed;
  * pixels are out of scope (the ray-cast renderer is not on the hot path): ``from_pixels=True`` raises.

Like `step`, the masked reset exists only on a HIP device (`reset_where` raises otherwise).
"""

from __future__ import annotations

import os
import re
from abc import abstractmethod

import torch

from .. import device_put, make_data, mjcf, reset_where, step
from ._compat import Bounded, Composite, EnvBase, TensorDict, Unbounded  # noqa: F401

_MODEL_DIR = os.environ.get(
    "MJH_MODEL_DIR", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "test_data")
)

ENVS: dict = {}


def register_env(name: str):
    """Class decorator: ``ENVS[name] = cls`` (reference zoo/base.py:19-26)."""

    def deco(cls):
        ENVS[name] = cls
        return cls

    return deco


_LIGHT_XML = '<light name="top" pos="0 0 4" dir="0 0 -1" diffuse="0.8 0.8 0.8" ambient="0.3 0.3 0.3" directional="true"/>'
_FLOOR_XML = '\n  <geom name="floor" type="plane" size="10 10 0.1" rgba="0.8 0.85 0.8 1" conaffinity="1" condim="3"/>'


class MujocoTorchEnv(EnvBase):
    """Batched environment; subclasses provide the XML name, observation, reward and termination."""

    RESET_NOISE_SCALE = 0.01
    FRAME_SKIP = 1
    ADD_FLOOR = True  # inject a ground plane when the model has none (zoo/base.py:249-254)
    CARRIED_INPUTS = ("ctrl",)  # input leaves the env edits between steps: copied into the spare buffer before a frame-skip loop

    def __init__(self, num_envs: int = 64, max_episode_steps: int = 1000, device=None, dtype=torch.float64,
                 compile_step: bool = False, compile_kwargs: dict | None = None, auto_reset: bool = False,
                 frame_skip: int | None = None, from_pixels: bool = False, pixel_only: bool = False,
                 render_width: int = 64, render_height: int = 64):
        if from_pixels or pixel_only:
            raise NotImplementedError("pixel observations need the ray-cast renderer, which is outside the stepper's scope")
        # compile_step / compile_kwargs are accepted for signature compatibility: there is nothing to compile, the step
        # is already a fixed sequence of native launches
        if frame_skip is not None:
            self.FRAME_SKIP = frame_skip
        self.auto_reset = auto_reset
        super().__init__(device=device, batch_size=torch.Size([num_envs]))
        self.dtype = dtype
        self.num_envs = num_envs
        self.max_episode_steps = max_episode_steps

        with open(os.path.join(_MODEL_DIR, self._xml_path())) as f:
            xml = self._patch_xml(f.read())
        self._xml = xml
        self._m_mj = mjcf.from_xml_string(xml, base_dir=_MODEL_DIR)
        self._dt = float(self._m_mj.opt.timestep) * self.FRAME_SKIP
        self.mx = device_put(self._m_mj)
        if device is not None:
            self.mx = self.mx.to(device)
        nu = int(self._m_mj.nu)

        self.observation_spec = Composite(**self._obs_spec_dict(num_envs, dtype, self.device), batch_size=[num_envs])
        low, high = self._action_range()
        self.action_spec = Bounded(low=low, high=high, shape=(num_envs, nu), dtype=dtype, device=self.device)
        self.reward_spec = Unbounded(shape=(num_envs, 1), dtype=dtype, device=self.device)

        # the reference seeds dx0 from mj_forward'ed MjData and runs one step (zoo/base.py:128-135); the stepper's own
        # make_data holds the same qpos0 state
        dx0 = make_data(self.mx)
        if device is not None:
            dx0 = dx0.to(device)
        self._dx0 = step(self.mx, dx0)
        self._sim_dtype = self._dx0.qpos.dtype
        self._ctrl_dtype = self._dx0.ctrl.dtype
        self._single_env = False  # one environment is the B = 1 batch of the same launches
        self._spare = None
        self._physics_step = self._native_multi_step

    # ---- subclass interface (zoo/base.py:164-205) -------------------------------------------------------------------

    @classmethod
    @abstractmethod
    def _xml_path(cls) -> str: ...

    @staticmethod
    @abstractmethod
    def _obs_spec_dict(num_envs: int, dtype: torch.dtype, device: torch.device) -> dict: ...

    @abstractmethod
    def _make_obs(self) -> dict: ...

    @abstractmethod
    def _compute_reward(self, qpos_before: torch.Tensor, action: torch.Tensor) -> torch.Tensor: ...

    @abstractmethod
    def _compute_terminated(self) -> torch.Tensor: ...

    @classmethod
    def _action_range(cls):
        return (-1.0, 1.0)

    def _prepare_ctrl(self, action: torch.Tensor) -> torch.Tensor:
        return action.to(self._ctrl_dtype)

    def _build_obs(self) -> dict:
        return self._make_obs()

    @classmethod
    def _camera_xml(cls) -> str:
        return '<camera name="side" pos="0 -4 3" xyaxes="1 0 0 0 0.45 1" fovy="60"/>'

    @classmethod
    def _patch_xml(cls, xml: str) -> str:
        """One fixed camera and light instead of the model's, plus a floor if there is no plane (zoo/base.py:235-264)."""
        xml = re.sub(r"<camera\b[^/]*/>\s*", "", xml)
        xml = re.sub(r"<light\b[^/]*/>\s*", "", xml)
        floor = _FLOOR_XML if cls.ADD_FLOOR and not re.search(r'<geom\b[^>]*type="plane"', xml) else ""
        return xml.replace("<worldbody>", f"<worldbody>\n  {cls._camera_xml()}\n  {_LIGHT_XML}{floor}")

    # ---- physics ----------------------------------------------------------------------------------------------------

    def _native_multi_step(self, d):
        """``frame_skip`` steps over two resident buffers; returns the buffer that holds the final state."""
        if self._spare is None or self._spare.qpos.shape != d.qpos.shape:
            self._spare = d.clone()
        cur, other = d, self._spare
        for name in self.CARRIED_INPUTS:  # a step does not write its inputs: the second buffer needs them too
            getattr(other, name).copy_(getattr(cur, name))
        for _ in range(self.FRAME_SKIP):
            step(self.mx, cur, out=other)
            cur, other = other, cur
        self._spare = other
        return cur

    def _reset_state(self, n: int):
        """qpos / qvel of n fresh environments: dx0 + U(-noise, noise) (zoo/base.py:266-273).  Override to edit them."""
        q0 = self._dx0.qpos.reshape(1, -1).expand(n, -1).clone()
        v0 = self._dx0.qvel.reshape(1, -1).expand(n, -1).clone()
        noise = self.RESET_NOISE_SCALE
        if noise > 0:
            q0.add_(torch.empty_like(q0).uniform_(-noise, noise))
            v0.add_(torch.empty_like(v0).uniform_(-noise, noise))
        return q0, v0

    def _make_batch(self, n: int):
        batch = self._dx0.expand(n).clone()
        q, v = self._reset_state(n)
        batch.qpos.copy_(q)
        batch.qvel.copy_(v)
        return batch

    def _reset_masked(self, mask: torch.Tensor):
        """``self._dx[mask] = self._make_batch(n)``; ``self._step_count[mask] = 0`` -- one native launch, no host sync."""
        q, v = self._reset_state(self.num_envs)
        reset_where(self.mx, self._dx, self._dx0, mask, q, v)  # raises off-device: there is no CPU route
        self._step_count.masked_fill_(mask, 0)

    # ---- TorchRL interface --------------------------------------------------------------------------------------------

    def _flags(self, value: bool = False):
        """A (*batch, 1) bool tensor: TorchRL's layout for done / terminated / truncated."""
        return torch.full((*self.batch_size, 1), value, dtype=torch.bool, device=self.device)

    def _start_episodes(self):
        """Every environment from scratch: fresh batch, step counters at zero."""
        self._dx = self._make_batch(self.num_envs)
        self._step_count = torch.zeros(self.num_envs, dtype=torch.long, device=self.device)

    def _upload_ctrl(self, ctrl):
        """The controls of this step into the resident Data.  Written in place when the leaf has the right layout, so the device pointers
        the native step cached for the ping-pong buffers stay valid; a differently shaped / typed control replaces the leaf."""
        leaf = self._dx.ctrl
        if (leaf.shape, leaf.dtype, leaf.device) == (ctrl.shape, ctrl.dtype, ctrl.device):
            leaf.copy_(ctrl)
        else:
            self._dx.update_(ctrl=ctrl)

    def _reset(self, tensordict=None, **kwargs):
        # TorchRL hands a partial reset as tensordict["_reset"]; anything else (first call, reset() without a mask) restarts all
        partial = tensordict["_reset"].squeeze(-1) if (tensordict is not None and "_reset" in tensordict.keys()) else None
        if partial is None or not hasattr(self, "_dx"):
            self._start_episodes()
        elif not self.auto_reset:
            self._reset_masked(partial)  # (with auto_reset on, _step has already replaced the finished environments)
        out = dict(self._build_obs())
        out["done"] = self._flags()
        out["terminated"] = self._flags()
        return TensorDict(out, batch_size=self.batch_size, device=self.device)

    def _step(self, tensordict):
        action = tensordict["action"].to(self.dtype)
        self._upload_ctrl(self._prepare_ctrl(action))
        before = self._dx.qpos.clone()          # rewards are functions of the displacement over the frame-skipped step
        self._dx = self._physics_step(self._dx)
        self._step_count += 1

        out = {"reward": self._compute_reward(before, action)}
        out["terminated"] = self._compute_terminated()
        out["done"] = out["terminated"] | (self._step_count >= self.max_episode_steps).unsqueeze(-1)
        out.update(self._build_obs())           # the observation of the state that earned the reward: taken BEFORE any reset below
        if self.auto_reset:
            self._reset_masked(out["done"].squeeze(-1))  # one masked native launch; no host sync on done.any()
        return TensorDict(out, batch_size=self.batch_size, device=self.device)

    def _set_seed(self, seed):
        torch.manual_seed(seed)
