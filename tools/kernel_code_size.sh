#!/bin/bash
# Code size (bytes) of every kernel in the gfx950 code object: the solver loop has to fit the 64 KB instruction cache.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
TMP="$(mktemp -d)"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off --cuda-device-only -c "$HERE/../mujoco-torch_amd/csrc/mjhip.hip" -o "$TMP/dev.o" "$@"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$TMP/dev.o" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$TMP/dev.co"
/opt/rocm/lib/llvm/bin/llvm-readelf -s "$TMP/dev.co" | awk '$4=="FUNC"{print $3, $8}' | sort -u | sort -n
rm -rf "$TMP"
