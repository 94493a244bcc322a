"""Worker for tests/test_multiprocess.py: each rank steps its shard with the CPU oracle (there is no GPU in
the CPU test tier), then the state is gathered and rank 0 compares with the unsharded result."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mujoco-torch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch
import torch.distributed as dist

import mujoco_torch_amd as mt
import pyoracle
from _util import load_model
from mujoco_torch_amd.sharding import gather_state, shard_data, shard_range


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    B = 11  # deliberately not divisible by the world size
    mx = load_model("humanoid", {"solver": 1})
    d = mt.make_data(mx).expand(B).clone().replace(qvel=torch.tensor(0.01 * np.random.RandomState(5).randn(B, mx.nv)))
    mine = shard_data(d, world, rank)
    b, e = shard_range(B, world, rank)
    assert mine.qpos.shape[0] == e - b
    for _ in range(2):
        mine = pyoracle.apply(mine, pyoracle.run(mx, mine, step=True))
    g = gather_state(mine, ("qpos", "qvel", "qacc"))
    if rank == 0:
        full = d
        for _ in range(2):
            full = pyoracle.apply(full, pyoracle.run(mx, full, step=True))
        for n in ("qpos", "qvel", "qacc"):
            assert torch.equal(g[n], getattr(full, n)), n
        print("MP_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
