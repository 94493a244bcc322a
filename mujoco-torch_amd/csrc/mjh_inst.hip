// mjh_inst.hip -- one build group of kernel instantiations (mjh_instances.h).  Compiled once per (group, dtype) by build.sh:
//   hipcc -c -DMJH_INST_GROUP=<g> -DMJH_INST_REAL=<double|float> mjh_inst.hip
#include <hip/hip_runtime.h>

#include "mjh_kernels.h"
#include "mjh_convex.h"
#include "mjh_sensor.h"
#include "mjh_instances.h"

#define MJH_CAT_(a, b) a##b
#define MJH_CAT(a, b) MJH_CAT_(a, b)
#define X_(R, P, W) template __global__ void mjh_phase_kernel<R, P, W>(KArgs<R>);
#define S_(R, N, RPL, W) template __global__ void mjh_sol2_kernel<R, N, RPL, W>(KArgs<R>);
#define C_(R) template __global__ void mjh_convex_kernel<R>(KArgs<R>);
#define N_(R) template __global__ void mjh_sensor_kernel<R, 0>(KArgs<R>); template __global__ void mjh_sensor_kernel<R, 1>(KArgs<R>);
MJH_CAT(MJH_INST_G, MJH_INST_GROUP)(X_, S_, C_, N_, MJH_INST_REAL)
