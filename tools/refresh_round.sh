#!/bin/bash
# GPU box: regenerates every measured artefact of a round under gpurun_out/<round>/ (copy to profiles/<round>/ afterwards):
# rocprofv3 kernel stats, PMC traffic (stamped with the library fingerprint bench.py checks), parity report, counter calibration,
# and the bench lines of the three BASELINE workloads.  Usage: MJH_GIT_COMMIT=<sha> bash tools/refresh_round.sh r03
R=${1:-r06}
O=gpurun_out/$R
mkdir -p $O
uptime > $O/host_load.txt  # the boxes' hosts are shared: a loaded host starves the calling thread of the drop-in loop (profiles/r03/notes.md)
bash tools/fetch_calib.sh 2>&1 | grep -E "FETCH_SIZE|WRITE_SIZE" > $O/fetch_calibration.txt
for w in humanoid ant mesh; do
  bash tools/prof_kernels.sh $w 200 > $O/kernel_stats_$w.txt 2>&1
  cp gpurun_out/${w}_kernel_stats.csv $O/${w}_kernel_stats.csv
  bash tools/hbm_traffic.sh $w > /dev/null 2>&1
done
python tools/parity_report.py > $O/parity_report.log 2>&1
cp gpurun_out/parity.json $O/parity.json
# traffic + parity files must be where bench.py looks for them before the bench lines are taken
mkdir -p profiles/$R
python - $R <<'PY'
import json, shutil, sys
R = sys.argv[1]
names = {"humanoid": "humanoid_b4096_f64", "ant": "ant_b16384_f32", "mesh": "mesh_b8192_f32"}
for w, tag in names.items():
    shutil.copy(f"gpurun_out/hbm_traffic_{w}.json", f"profiles/{R}/hbm_traffic_{tag}.json")
    shutil.copy(f"gpurun_out/hbm_traffic_{w}.json", f"gpurun_out/{R}/hbm_traffic_{tag}.json")
shutil.copy("gpurun_out/parity.json", f"profiles/{R}/parity.json")
PY
python bench.py --steps 200 --warmup 20 > $O/bench_humanoid.json 2> $O/bench_humanoid.err   # carries ant + mesh as other_workloads
python bench.py --steps 20 --warmup 5 > $O/bench_humanoid_driver_flags.json 2> /dev/null            # the driver's round-end flags
for w in ant mesh; do
  python bench.py --workload $w --steps 200 --warmup 20 > $O/bench_$w.json 2> $O/bench_$w.err
done
python bench.py --workload humanoid32k --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_humanoid32k.json 2>/dev/null
MJH_BENCH_SHARE_GPU=1 MJH_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_humanoid_2ranks_one_gpu.json 2>/dev/null
python - $R <<'PY'
import json, sys
R = sys.argv[1]
for w in ("humanoid", "ant", "mesh", "humanoid32k"):
    j = json.load(open(f"gpurun_out/{R}/bench_{w}.json"))
    r = j["roofline"]
    print(f"{w:12s} {j['value'] / 1e6:7.3f} M env-steps/s  {j['ms_per_step']:.4f} ms/step  out= {j['out_buffers']['value'] / 1e6:7.3f} M  dominant {r['kernel'][:44]} {r['kernel_avg_us']:.1f} us frac {r['frac']:.3f} traffic {r['traffic']}  step frac {r['step']['frac']:.3f} step traffic {r['step']['traffic']}")
PY
