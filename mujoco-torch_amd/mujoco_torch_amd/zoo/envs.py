"""The reference's environment zoo (reference zoo/{ant,cartpole,halfcheetah,hopper,humanoid,humanoid_rich,swimmer,walker2d,
satellite}.py) restated on one declarative class: observation slices, the forward-velocity / healthy / control-cost reward
and the healthy-range termination are class attributes, so an environment is a table row rather than a module.

Class names, registry keys, observation layouts, reward terms and constants are the reference's (file:line per class).
"""

from __future__ import annotations

import re

import torch

from ._compat import Bounded, Unbounded
from .base import MujocoTorchEnv, register_env


class LocomotionEnv(MujocoTorchEnv):
    """observation = [qpos[skip_q:], (clipped) qvel[skip_v:]]; reward = forward velocity + healthy bonus - ctrl cost."""

    XML = ""
    OBS_DIM = 0
    SKIP_QPOS = 0          # leading qpos entries left out of the observation
    SKIP_QVEL = 0
    CLIP_QVEL = None       # clamp qvel to +-this in the observation
    FORWARD_WEIGHT = 1.0   # weight of (x_after - x_before) / dt, x = qpos[0]
    CTRL_COST_WEIGHT = 0.0
    HEALTHY_REWARD = 0.0
    HEALTHY_Z = None       # (qpos index, low, high): outside -> unhealthy -> terminated
    HEALTHY_ANGLE = None   # (qpos index, max |angle|)
    CONSTANT_REWARD = None # reward independent of the state (cartpole)

    @classmethod
    def _xml_path(cls):
        return cls.XML

    @classmethod
    def _obs_spec_dict(cls, num_envs, dtype, device):
        return {"observation": Unbounded(shape=(num_envs, cls.OBS_DIM), dtype=dtype, device=device)}

    def _make_obs(self):
        qpos = self._dx.qpos.to(self.dtype)
        qvel = self._dx.qvel.to(self.dtype)
        if self.CLIP_QVEL is not None:
            qvel = qvel.clamp(-self.CLIP_QVEL, self.CLIP_QVEL)
        return {"observation": torch.cat([qpos[..., self.SKIP_QPOS:], qvel[..., self.SKIP_QVEL:]], dim=-1)}

    def _is_healthy(self):
        q = self._dx.qpos
        ok = torch.ones(q.shape[:-1], dtype=torch.bool, device=q.device)
        if self.HEALTHY_Z is not None:
            i, lo, hi = self.HEALTHY_Z
            ok = ok & (q[..., i] >= lo) & (q[..., i] <= hi)
        if self.HEALTHY_ANGLE is not None:
            i, amax = self.HEALTHY_ANGLE
            ok = ok & (q[..., i].abs() <= amax)
        return ok

    def _compute_reward(self, qpos_before, action):
        if self.CONSTANT_REWARD is not None:
            return torch.full((*self.batch_size, 1), self.CONSTANT_REWARD, dtype=self.dtype, device=self.device)
        reward = self.FORWARD_WEIGHT * (self._dx.qpos[..., 0] - qpos_before[..., 0]) / self._dt
        if self.HEALTHY_REWARD:
            reward = reward + torch.where(self._is_healthy(), self.HEALTHY_REWARD, 0.0)
        reward = reward - self.CTRL_COST_WEIGHT * (action**2).sum(dim=-1)
        return reward.unsqueeze(-1).to(self.dtype)

    def _compute_terminated(self):
        if self.HEALTHY_Z is None and self.HEALTHY_ANGLE is None:
            return torch.zeros(*self.batch_size, 1, dtype=torch.bool, device=self.device)
        return (~self._is_healthy()).unsqueeze(-1)


@register_env("ant")
class AntEnv(LocomotionEnv):
    """Gymnasium Ant-v4 on the bundled fixed-base ant.xml: a free joint is inserted on the torso and the timestep set to
    0.01 (zoo/ant.py:23-80).  nq 15, nv 14, nu 8; observation qpos[2:] + qvel = 27."""

    XML, OBS_DIM, SKIP_QPOS = "ant.xml", 27, 2
    RESET_NOISE_SCALE, FRAME_SKIP = 0.1, 5
    HEALTHY_Z, HEALTHY_REWARD, CTRL_COST_WEIGHT = (2, 0.2, 1.0), 1.0, 0.5

    @classmethod
    def _patch_xml(cls, xml):
        xml = super()._patch_xml(xml)
        xml = re.sub(r'(<body\s+name="torso"[^>]*>)', r"\1\n      <freejoint name='root'/>", xml, count=1)
        return re.sub(r"(<compiler\b[^/]*/>\s*)", r'\1<option timestep="0.01"/>\n  ', xml, count=1)


@register_env("cartpole")
class CartPoleEnv(LocomotionEnv):
    """zoo/cartpole.py:14-45: reward 1 per step, terminated when |pole angle| > 0.2; observation qpos + qvel = 4."""

    XML, OBS_DIM = "cartpole.xml", 4
    CONSTANT_REWARD = 1.0
    ANGLE_LIMIT = 0.2

    @classmethod
    def _camera_xml(cls):
        return '<camera name="side" pos="0 -2 1.5" xyaxes="1 0 0 0 0.45 1" fovy="60"/>'

    def _compute_terminated(self):
        return (self._dx.qpos[..., 1].abs() > self.ANGLE_LIMIT).unsqueeze(-1)


@register_env("halfcheetah")
class HalfCheetahEnv(LocomotionEnv):
    """zoo/halfcheetah.py:14-42: forward velocity - 0.1 * |action|^2, never terminates; observation qpos[1:] + qvel = 17."""

    XML, OBS_DIM, SKIP_QPOS = "halfcheetah.xml", 17, 1
    RESET_NOISE_SCALE, FRAME_SKIP = 0.1, 5
    CTRL_COST_WEIGHT = 0.1


@register_env("hopper")
class HopperEnv(LocomotionEnv):
    """zoo/hopper.py:14-53: healthy while rootz >= 0.7 and |rooty| <= 0.2; observation qpos[1:] + clipped qvel = 11."""

    XML, OBS_DIM, SKIP_QPOS, CLIP_QVEL = "hopper.xml", 11, 1, 10.0
    HEALTHY_Z_MIN, HEALTHY_ANGLE_MAX = 0.7, 0.2
    HEALTHY_Z, HEALTHY_ANGLE = (1, 0.7, float("inf")), (2, 0.2)
    HEALTHY_REWARD, CTRL_COST_WEIGHT = 1.0, 1e-3


@register_env("walker2d")
class Walker2dEnv(LocomotionEnv):
    """zoo/walker2d.py:14-54: healthy while 0.8 <= rootz <= 2.0 and |rooty| <= 1.0; observation qpos[1:] + clipped qvel = 17."""

    XML, OBS_DIM, SKIP_QPOS, CLIP_QVEL = "walker2d.xml", 17, 1, 10.0
    HEALTHY_Z, HEALTHY_ANGLE = (1, 0.8, 2.0), (2, 1.0)
    HEALTHY_REWARD, CTRL_COST_WEIGHT = 1.0, 1e-3


@register_env("humanoid")
class HumanoidEnv(LocomotionEnv):
    """zoo/humanoid.py:16-59: healthy while 1.0 <= z <= 2.0; observation qpos[2:] (26) + clipped qvel (27) = 53."""

    XML, OBS_DIM, SKIP_QPOS, CLIP_QVEL = "humanoid.xml", 53, 2, 10.0
    RESET_NOISE_SCALE, FRAME_SKIP = 0.01, 5
    HEALTHY_Z, HEALTHY_REWARD, CTRL_COST_WEIGHT = (2, 1.0, 2.0), 5.0, 0.1

    @classmethod
    def _camera_xml(cls):
        return '<camera name="side" pos="0 -6 3" xyaxes="1 0 0 0 0.45 1" fovy="60"/>'


@register_env("humanoid_rich")
class HumanoidRichEnv(HumanoidEnv):
    """zoo/humanoid_rich.py:31-58: Humanoid-v5 style observation, + cinert[1:] (160) + cvel[1:] (96) + qfrc_actuator (27) = 336."""

    OBS_DIM = 336

    def _make_obs(self):
        base = super()._make_obs()["observation"]
        d = self._dx
        extra = [d.cinert[..., 1:, :].to(self.dtype).flatten(-2), d.cvel[..., 1:, :].to(self.dtype).flatten(-2), d.qfrc_actuator.to(self.dtype)]
        return {"observation": torch.cat([base, *extra], dim=-1)}


@register_env("swimmer")
class SwimmerEnv(LocomotionEnv):
    """zoo/swimmer.py:17-57: forward velocity - 1e-4 * |action|^2, never terminates; observation qpos[2:] (7) + qvel[2:] (6) = 13."""

    XML, OBS_DIM, SKIP_QPOS, SKIP_QVEL = "swimmer.xml", 13, 2, 2
    CTRL_COST_WEIGHT = 1e-4


class _SatelliteBase(MujocoTorchEnv):
    """Attitude control with control-moment gyros (zoo/satellite.py:32-131): the agent commands the gimbal rates, the rotor
    speed actuators are held at ROTOR_SPEED; reward = sun alignment of body +Z - control cost - angular-velocity penalty."""

    N_GIMBALS = 0
    ROTOR_SPEED = 100.0
    FRAME_SKIP = 10
    RESET_NOISE_SCALE = 0.001
    CTRL_COST_WEIGHT = 0.01
    ANG_VEL_WEIGHT = 0.1
    ADD_FLOOR = False  # no ground in orbit (zoo/satellite.py:52-67)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.action_spec = Bounded(low=-1.0, high=1.0, shape=(self.num_envs, self.N_GIMBALS), dtype=self.dtype, device=self.device)

    @classmethod
    def _obs_spec_dict(cls, num_envs, dtype, device):
        return {"observation": Unbounded(shape=(num_envs, 7 + 4 * cls.N_GIMBALS), dtype=dtype, device=device)}

    def _make_obs(self):
        qpos, qvel = self._dx.qpos.to(self.dtype), self._dx.qvel.to(self.dtype)
        # bus quaternion, bus angular velocity, gimbal / rotor angles, gimbal / rotor rates
        return {"observation": torch.cat([qpos[..., 3:7], qvel[..., 3:6], qpos[..., 7:], qvel[..., 6:]], dim=-1)}

    def _prepare_ctrl(self, action):
        rotors = torch.full((*self.batch_size, self.N_GIMBALS), self.ROTOR_SPEED, dtype=self._ctrl_dtype, device=self.device)
        return torch.cat([action.to(self._ctrl_dtype), rotors], dim=-1)

    def _reset_state(self, n):
        q, v = super()._reset_state(n)
        v[..., [7 + 2 * i for i in range(self.N_GIMBALS)]] = self.ROTOR_SPEED  # rotors start spun up
        return q, v

    def _compute_reward(self, qpos_before, action):
        qx, qy = self._dx.qpos[..., 4], self._dx.qpos[..., 5]
        sun = 1.0 - 2.0 * (qx**2 + qy**2)  # z component of the body +Z axis in the world frame
        spin = self._dx.qvel[..., 3:6]
        reward = sun - self.CTRL_COST_WEIGHT * (action**2).sum(dim=-1) - self.ANG_VEL_WEIGHT * (spin**2).sum(dim=-1)
        return reward.unsqueeze(-1).to(self.dtype)

    def _compute_terminated(self):
        return torch.zeros(*self.batch_size, 1, dtype=torch.bool, device=self.device)


@register_env("satellite_large")
class SatelliteLargeEnv(_SatelliteBase):
    """4 CMGs in a pyramid; nq 15, nv 14, nu 8 (zoo/satellite.py:134-147)."""

    N_GIMBALS, ROTOR_SPEED = 4, 100.0

    @classmethod
    def _xml_path(cls):
        return "satellite_large.xml"

    @classmethod
    def _camera_xml(cls):
        return '<camera name="side" pos="3 -3 2" xyaxes="0.707 0.707 0 -0.302 0.302 0.905" fovy="60"/>'


@register_env("satellite_small")
class SatelliteSmallEnv(_SatelliteBase):
    """CubeSat with 6 CMGs; nq 19, nv 18, nu 12 (zoo/satellite.py:150-162)."""

    N_GIMBALS, ROTOR_SPEED = 6, 200.0

    @classmethod
    def _xml_path(cls):
        return "satellite_small.xml"

    @classmethod
    def _camera_xml(cls):
        return '<camera name="side" pos="0.5 -0.5 0.3" xyaxes="0.707 0.707 0 -0.276 0.276 0.920" fovy="60"/>'
