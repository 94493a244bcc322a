"""GPU box: agent-steps/s of a zoo environment with fused auto-reset, native masked reset vs the reference's route.

    python tools/env_throughput.py [env] [num_envs] [agent_steps]

"reference route" = `if mask.any(): self._dx[mask] = self._make_batch(int(mask.sum()))` (zoo/base.py:327-331) on the same
device tensors: a host sync for the count, a gather of n fresh environments and one index_put per leaf.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mujoco-torch_amd"))
import torch  # noqa: E402

from mujoco_torch_amd.zoo import ENVS  # noqa: E402
from mujoco_torch_amd.zoo._compat import TensorDict  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "humanoid"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 60


def bench(route):
    torch.manual_seed(0)
    env = ENVS[name](num_envs=B, device="cuda", auto_reset=True, max_episode_steps=25)
    if route == "index_put":
        def reset_masked(mask):
            if mask.any():
                n = int(mask.sum())
                env._dx[mask] = env._make_batch(n)
                env._step_count[mask] = 0
        env._reset_masked = reset_masked
    env.reset()
    env._step_count.copy_(torch.randint(0, 25, (B,), device="cuda"))  # staggered episodes: ~4 % of the envs reset per step
    nu = env.action_spec.shape[-1]
    acts = [torch.empty(B, nu, dtype=env.dtype, device="cuda").uniform_(-0.3, 0.3) for _ in range(8)]
    nreset = 0
    for i in range(T + 10):
        if i == 10:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = env._step(TensorDict({"action": acts[i % 8]}, batch_size=env.batch_size))
        if i >= 10:
            nreset += out["done"].sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(route=route, env=name, num_envs=B, frame_skip=env.FRAME_SKIP, agent_steps_per_s=B * T / dt, env_steps_per_s=B * T * env.FRAME_SKIP / dt,
                ms_per_agent_step=1e3 * dt / T, resets_per_agent_step=float(nreset) / T)


for route in ("native", "index_put"):
    print(json.dumps(bench(route)))
