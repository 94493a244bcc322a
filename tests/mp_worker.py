"""Worker for the 2-rank tests: each rank steps its contiguous shard of one seeded batch THROUGH THE PRODUCT (`mt.step`), the
state is gathered (gloo) and rank 0 compares with the unsharded step, bit for bit.

    MJH_MP_DEVICE=cuda : both ranks share cuda:0 and run the HIP library (tests/test_gpu_parity.py, -m gpu)
    MJH_MP_DEVICE=cpu  : no GPU in the CPU tier -- the product's host path runs against tests/_hostsim.py (the oracle answers the
                         library calls), which still covers sharding + the per-rank host logic under torch.distributed
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mujoco-torch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch
import torch.distributed as dist

import mujoco_torch_amd as mt
from _util import load_model
from mujoco_torch_amd.sharding import gather_state, shard_data, shard_range


class _Patch:  # minimal monkeypatch stand-in for _hostsim.install outside pytest
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def main():
    device = os.environ.get("MJH_MP_DEVICE", "cpu")
    if device == "cpu":
        import _hostsim

        _hostsim.install(_Patch())
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    B = 11  # deliberately not divisible by the world size
    names = ("qpos", "qvel", "qacc", "efc_force", "time")
    for xml, ov, dt in (("humanoid", {"solver": 1}, torch.float64), ("ant", {"integrator": 1, "solver": 2, "cone": 1}, torch.float32)):
        mx = load_model(xml, ov, dt)
        d = mt.make_data(mx).expand(B).clone().replace(qvel=torch.tensor(0.01 * np.random.RandomState(5).randn(B, mx.nv)))
        if dt != torch.float64:
            d = d.to(dt)
        mdev = mx.to(device)
        mine = shard_data(d, world, rank).clone().to(device)
        b, e = shard_range(B, world, rank)
        assert mine.qpos.shape[0] == e - b
        for _ in range(3):
            mine = mt.step(mdev, mine)
        g = gather_state(mine.to("cpu"), names)
        if rank == 0:
            full = d.to(device)
            for _ in range(3):
                full = mt.step(mdev, full)
            for n in names:
                assert torch.equal(g[n], getattr(full, n).cpu()), (xml, n)
    if rank == 0:
        print("MP_OK")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
