// mjh_instances.h -- the list of kernel instantiations the library ships, cut into build groups.
//
// Every kernel is one fully inlined function of several thousand lines; compiled in one translation unit the library took four minutes.
// Each group below is compiled by its own hipcc process (mjh_inst.hip with -DMJH_INST_GROUP=g -DMJH_INST_REAL=double|float, build.sh runs
// them in parallel); mjhip.hip sees the same list as `extern template` declarations, so it holds the host side only.
// X(REAL, PHASE, W) for mjh_phase_kernel, S(REAL, NMAX, RPL, W) for mjh_sol2_kernel, C(REAL) / N(REAL) for the convex and sensor kernels.
#pragma once

#define MJH_INST_G0(X, S, C, N, R) X(R, 0, 64) X(R, 0, 32) X(R, 0, 16)
#define MJH_INST_G1(X, S, C, N, R) X(R, 1, 64) X(R, 1, 32) X(R, 1, 16) X(R, 2, 64)
#define MJH_INST_G2(X, S, C, N, R) X(R, 3, 64) X(R, 3, 32) X(R, 3, 16)
#define MJH_INST_G3(X, S, C, N, R) X(R, 12, 64) X(R, 12, 32) X(R, 12, 16)
#define MJH_INST_G4(X, S, C, N, R) X(R, 5, 64) X(R, 5, 32) X(R, 5, 16)
#define MJH_INST_G5(X, S, C, N, R) X(R, 4, 64) X(R, 6, 64) X(R, 7, 64)
#define MJH_INST_G6(X, S, C, N, R) X(R, 8, 64) X(R, 8, 32) C(R) N(R)
#define MJH_INST_G7(X, S, C, N, R) S(R, 8, 1, 32) S(R, 8, 2, 32) S(R, 8, 4, 32) S(R, 8, 8, 32)
#define MJH_INST_G8(X, S, C, N, R) S(R, 16, 1, 32) S(R, 16, 2, 32) S(R, 16, 4, 32) S(R, 16, 8, 32)
#define MJH_INST_G9(X, S, C, N, R) S(R, 28, 1, 32) S(R, 28, 2, 32)
#define MJH_INST_G10(X, S, C, N, R) S(R, 8, 2, 16) S(R, 8, 5, 16) S(R, 12, 2, 16) S(R, 12, 5, 16)  /* four environments per wavefront (nv <= 16) */
#define MJH_INST_G11(X, S, C, N, R) S(R, 16, 2, 16) S(R, 16, 5, 16)
#define MJH_INST_G13(X, S, C, N, R) S(R, 8, 2, 17) S(R, 8, 5, 17) S(R, 12, 2, 17) S(R, 12, 5, 17)  /* ... the same for Newton models (W = 17: 16 lanes, Newton-only code) */
#define MJH_INST_G14(X, S, C, N, R) S(R, 16, 2, 17) S(R, 16, 5, 17)
#define MJH_INST_G12(X, S, C, N, R) X(R, 13, 64) X(R, 13, 32) X(R, 13, 16)  /* kinematics + crb + velocity in one launch */
#define MJH_INST_G15(X, S, C, N, R) X(R, 2, 32)  /* plain constraint phase, two environments per wavefront (mid-size models) */
#define MJH_INST_G16(X, S, C, N, R) S(R, 28, 1, 33) S(R, 28, 1, 35)  /* constraint stage + register solver + integrator in one kernel (W = 33: 32 lanes, fused; W = 35: the same for opt.iterations == 1) */
#define MJH_INST_G17(X, S, C, N, R) S(R, 28, 1, 34) S(R, 28, 1, 36)  /* the whole pass -- kinematics + crb + velocity + constraint stage + register solver + integrator -- in one kernel (W = 34) */
#define MJH_INST_G18(X, S, C, N, R) X(R, 17, 32) X(R, 17, 16)  /* kernel 13 on two wavefronts per workgroup: kinematics, then velocity beside crb / factor */
#define MJH_INST_G19(X, S, C, N, R) S(R, 8, 2, 18) S(R, 8, 5, 18)  /* one RK4 stage of a small Newton model in one launch: kernel 13's stages + constraint phase + the solver's first tier (W = 18) */
#define MJH_INST_G20(X, S, C, N, R) S(R, 12, 2, 18) S(R, 12, 5, 18)
#define MJH_INST_G21(X, S, C, N, R) S(R, 16, 2, 18) S(R, 16, 5, 18)
#define MJH_INST_NGROUPS 22

#define MJH_INST_ALL(X, S, C, N, R)                                                                                              \
  MJH_INST_G0(X, S, C, N, R) MJH_INST_G1(X, S, C, N, R) MJH_INST_G2(X, S, C, N, R) MJH_INST_G3(X, S, C, N, R) MJH_INST_G4(X, S, C, N, R) \
  MJH_INST_G5(X, S, C, N, R) MJH_INST_G6(X, S, C, N, R) MJH_INST_G7(X, S, C, N, R) MJH_INST_G8(X, S, C, N, R) MJH_INST_G9(X, S, C, N, R) MJH_INST_G10(X, S, C, N, R) MJH_INST_G11(X, S, C, N, R) MJH_INST_G12(X, S, C, N, R) MJH_INST_G13(X, S, C, N, R) MJH_INST_G14(X, S, C, N, R) MJH_INST_G15(X, S, C, N, R) MJH_INST_G16(X, S, C, N, R) MJH_INST_G17(X, S, C, N, R) MJH_INST_G18(X, S, C, N, R) MJH_INST_G19(X, S, C, N, R) MJH_INST_G20(X, S, C, N, R) MJH_INST_G21(X, S, C, N, R)
