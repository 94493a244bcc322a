"""GPU box: the driver's timed window (5 warm-up + 20 timed steps after the spin-up) step by step: device time of every step between events, the host's enqueue times, four windows in a row.
PREWARM=1 records the events once before the window (torch creates the HIP event at the first record()).   python tools/window_probe.py"""
import sys, time, os
R="/root/repo"
for p in ("", "tests", "mujoco-torch_amd"): sys.path.insert(0, os.path.join(R, p))
sys.argv=sys.argv[:1]
import torch, bench
import mujoco_torch_amd as mt
dev=torch.device("cuda:0")
wl, B, dtype, mx, mdev, loop = bench.setup_workload("humanoid", 0, dev, 0)
bench.spin_up(mdev, loop, 100)
loop.dropin(5)
for trial in range(4):
    torch.cuda.synchronize(dev)
    ms0 = torch.cuda.memory_stats(dev)
    evs=[torch.cuda.Event(enable_timing=True) for _ in range(21)]
    if os.environ.get('PREWARM'):
        for e in evs: e.record()
        torch.cuda.synchronize(dev)
    t0=time.perf_counter()
    evs[0].record()
    host=[]
    for i in range(20):
        loop.dropin(1); evs[i+1].record(); host.append(time.perf_counter())
    torch.cuda.synchronize(dev)
    t1=time.perf_counter()
    dt=[evs[i].elapsed_time(evs[i+1])*1e3 for i in range(20)]
    print(f"trial {trial}: wall {1e6*(t1-t0)/20:.1f} us/step; device per step:", " ".join(f"{x:.0f}" for x in dt))
    print("     host enqueue times (us since t0):", " ".join(f"{1e6*(h-t0):.0f}" for h in host[:8]))
