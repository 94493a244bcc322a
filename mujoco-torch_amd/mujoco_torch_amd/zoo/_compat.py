"""TorchRL / tensordict when they are installed, otherwise the few names the env classes need.

The reference's environments are ``torchrl.envs.EnvBase`` subclasses that return ``TensorDict`` objects
(reference zoo/base.py:1-12).  Neither package is a dependency of the stepper; when they are missing the stand-ins below
keep the same call shapes (``env.reset()``, ``env.step(td)``, ``env.rollout(n, policy)``, ``td["observation"]``,
``td["next", "reward"]``) on plain dicts so that the environment logic itself -- frame skip, masked reset, rewards -- is
usable and testable without them.
"""

from __future__ import annotations

import torch

try:  # pragma: no cover - not installed in the build image
    from tensordict import TensorDict
    from torchrl.data import Bounded, Composite, Unbounded
    from torchrl.envs import EnvBase

    HAVE_TORCHRL = True
except ImportError:
    HAVE_TORCHRL = False

    class TensorDict(dict):
        """dict of tensors with a batch size; nested access with tuple keys like the real one."""

        def __init__(self, source=None, batch_size=None, device=None):
            super().__init__(source or {})
            self.batch_size = torch.Size(batch_size if batch_size is not None else [])
            self.device = device

        def __getitem__(self, key):
            if isinstance(key, tuple):
                node = self
                for k in key:
                    node = dict.__getitem__(node, k)
                return node
            return dict.__getitem__(self, key)

        def get(self, key, default=None):
            try:
                return self[key]
            except KeyError:
                return default

        def set(self, key, value):
            self[key] = value
            return self

        def clone(self):
            return TensorDict({k: (v.clone() if hasattr(v, "clone") else v) for k, v in self.items()}, self.batch_size, self.device)

    class _Spec:
        def __init__(self, shape=(), dtype=torch.float32, device=None, low=None, high=None):
            self.shape, self.dtype, self.device, self.low, self.high = torch.Size(shape), dtype, device, low, high

        def rand(self):
            if self.low is None:
                return torch.randn(self.shape, dtype=self.dtype, device=self.device)
            return torch.empty(self.shape, dtype=self.dtype, device=self.device).uniform_(float(self.low), float(self.high))

    class Unbounded(_Spec):
        pass

    class Bounded(_Spec):
        def __init__(self, low, high, shape=(), dtype=torch.float32, device=None):
            super().__init__(shape, dtype, device, low, high)

    class Composite(dict):
        def __init__(self, batch_size=None, **specs):
            super().__init__(**specs)
            self.batch_size = torch.Size(batch_size or [])

    class EnvBase:
        """``reset`` / ``step`` / ``rollout`` drivers around the ``_reset`` / ``_step`` hooks (torchrl.envs.EnvBase shape)."""

        def __init__(self, device=None, batch_size=None):
            self.device = torch.device(device) if device is not None else torch.device("cpu")
            self.batch_size = torch.Size(batch_size or [])

        def set_seed(self, seed):
            self._set_seed(seed)
            return seed

        def reset(self, tensordict=None, **kwargs):
            return self._reset(tensordict, **kwargs)

        def rand_action(self, tensordict=None):
            td = tensordict if tensordict is not None else TensorDict({}, self.batch_size, self.device)
            td["action"] = self.action_spec.rand()
            return td

        def step(self, tensordict):
            nxt = self._step(tensordict)
            tensordict["next"] = nxt
            return tensordict

        def rollout(self, max_steps, policy=None, break_when_any_done=False):
            td = self.reset()
            frames = []
            for _ in range(max_steps):
                td = policy(td) if policy is not None else self.rand_action(td)
                td = self.step(td)
                frames.append(td)
                nxt = td["next"]
                if break_when_any_done and bool(nxt["done"].any()):
                    break
                td = TensorDict({k: v for k, v in nxt.items() if k not in ("reward",)}, self.batch_size, self.device)
            return frames
