#!/bin/bash
# round 4, experiment 1: humanoid f64 B=4096 -- baseline vs kernel 13 (KIN + CRB + VEL in one launch, needs the two-per-wavefront CRB stage)
O=gpurun_out/r04; mkdir -p $O
run() { tag=$1; shift; env "$@" python3 bench.py --workload humanoid --steps 200 --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag"; }
run base MJH_X=0
run kcv13 MJH_CRB_PACK=1
run crbpack_only MJH_CRB_PACK=1 MJH_FUSE_CRB=0
run graphs MJH_GRAPHS=1
run base MJH_X=0
