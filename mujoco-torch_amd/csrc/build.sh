#!/bin/bash
# Builds libmjhip.so for gfx950 in-tree (mujoco-torch_amd/lib/).  -ffp-contract=off: keep the reference's
# separate multiply/add rounding (no FMA contraction) so results track the float64 oracle to ~1e-15.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
mkdir -p "$HERE/../lib"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off "$@" \
  -o "$HERE/../lib/libmjhip.so" "$HERE/mjhip.hip"
