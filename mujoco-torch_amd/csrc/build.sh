#!/bin/bash
# Builds libmjhip.so for gfx950 in-tree (mujoco-torch_amd/lib/).
#  -ffp-contract=off : keep the reference's separate multiply/add rounding (no FMA contraction) so results
#                      track the float64 oracle to ~1e-15.
# The kernels are compiled as 44 translation units (mjh_instances.h: 22 groups x 2 dtypes) by parallel hipcc processes
# (MJH_BUILD_JOBS, default: the number of CPUs), objects cached under csrc/build/ by a hash of the sources and flags; mjhip.hip
# is the host side.  `build.sh -DFOO` passes extra flags to every compile.  MJH_BUILD_ONLY="3d 7f" rebuilds only those groups
# (d = double, f = float) and relinks with the cached rest -- for iterating on one kernel.
# Device code reads its launch parameters through the kernarg segment pointer, which is only valid inside
# the kernel function itself: the build FAILS if any device function was left un-inlined.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OBJ="${MJH_BUILD_DIR:-$HERE/build}"           # diagnostic builds (tools/stamps.py) keep their own objects and library name
OUT="${MJH_BUILD_OUT:-$HERE/../lib/libmjhip.so}"
mkdir -p "$HERE/../lib" "$OBJ"
HIPCC=/opt/rocm/bin/hipcc
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Rpass-analysis=kernel-resource-usage $*"
JOBS="${MJH_BUILD_JOBS:-$(nproc)}"
SRC_HASH=$(cat "$HERE"/*.h "$HERE"/mjh_inst.hip "$HERE/../../include/mjhip.h" | sha256sum | cut -c1-16)
FLAG_HASH=$(echo "$FLAGS" | sha256sum | cut -c1-8)
NG=$(grep -E "^#define MJH_INST_NGROUPS" "$HERE/mjh_instances.h" | awk '{print $3}')

compile_one() {  # $1 = group, $2 = d|f
  local g=$1 t=$2 real=double
  [ "$t" = f ] && real=float
  local o="$OBJ/inst_${g}${t}.o" stamp="$OBJ/inst_${g}${t}.stamp" log="$OBJ/inst_${g}${t}.log"
  if [ -f "$o" ] && [ "$(cat "$stamp" 2>/dev/null)" = "$SRC_HASH-$FLAG_HASH" ]; then return 0; fi
  if [ -n "$MJH_BUILD_ONLY" ] && [ -f "$o" ] && ! echo " $MJH_BUILD_ONLY " | grep -q " ${g}${t} "; then return 0; fi
  rm -f "$stamp"
  $HIPCC $FLAGS -c -DMJH_INST_GROUP=$g -DMJH_INST_REAL=$real -o "$o" "$HERE/mjh_inst.hip" 2> "$log" || { cat "$log" >&2; return 1; }
  echo "$SRC_HASH-$FLAG_HASH" > "$stamp"
}
export -f compile_one
export HERE OBJ HIPCC FLAGS SRC_HASH FLAG_HASH MJH_BUILD_ONLY

LIST=""
# the heavy groups (register solver: 7 8 9, fused kinematics + velocity: 3) start first
for g in 17 21 20 19 7 8 10 13 11 14 9 16 3 12 18 4 5 0 1 2 6 15; do [ "$g" -lt "$NG" ] && LIST="$LIST $g:d $g:f"; done
echo $LIST | tr ' ' '\n' | xargs -P "$JOBS" -I{} bash -c 'IFS=: read g t <<< "{}"; compile_one $g $t'
$HIPCC $FLAGS -c -o "$OBJ/mjhip.o" "$HERE/mjhip.hip" 2> "$OBJ/mjhip.log" || { cat "$OBJ/mjhip.log" >&2; exit 1; }
OBJS="$OBJ/mjhip.o"
for g in $(seq 0 $((NG - 1))); do OBJS="$OBJS $OBJ/inst_${g}d.o $OBJ/inst_${g}f.o"; done
$HIPCC --offload-arch=gfx950 -fPIC -shared -o "$OUT" $OBJS

LOG="$(mktemp)"
cat "$OBJ"/inst_*.log "$OBJ/mjhip.log" > "$LOG"
grep -E "error|warning: " "$LOG" || true
NFUNC=$(grep -c "Function Name:" "$LOG" || true)
NKERN=$(grep "Function Name:" "$LOG" | grep -cE "mjh_phase_kernel|mjh_sol2_kernel|mjh_convex_kernel|mjh_sensor_kernel|mjh_reset_kernel|mjh_sort_kernel" || true)
if [ "$NFUNC" != "$NKERN" ]; then
  echo "build.sh: device functions were not inlined into the kernels:" >&2
  grep "Function Name:" "$LOG" | grep -vE "mjh_phase_kernel|mjh_sol2_kernel|mjh_convex_kernel|mjh_sensor_kernel|mjh_reset_kernel|mjh_sort_kernel" >&2
  exit 1
fi
grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" "$LOG" | sed 's/.*remark: *//' | paste - - - - | sed 's/\[-Rpass[^]]*\]//g' | sort > "${OUT%.so}.resource_usage.txt"
[ "$OUT" = "$HERE/../lib/libmjhip.so" ] && mv "${OUT%.so}.resource_usage.txt" "$HERE/../lib/resource_usage.txt"
rm -f "$LOG"
