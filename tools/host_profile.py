"""GPU box: where the host time of `d = step(mx, d)` goes (cProfile over calls at B = 2: the device time is negligible)."""
import cProfile, os, pstats, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "mujoco-torch_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import torch
import mujoco_torch_amd as mt
from _util import load_model
mx = load_model("humanoid", {"solver": 1}, torch.float64)
mdev = mx.to("cuda")
d = mt.make_data(mx).expand(2).clone().to("cuda")
for _ in range(200): d = mt.step(mdev, d)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3000): d = mt.step(mdev, d)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
