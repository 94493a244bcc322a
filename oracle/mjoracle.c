/*
 * mjoracle.c -- CPU oracle for the hot path: mjo_forward / mjo_step (twins of mjh_forward / mjh_step).
 *
 * TEST INFRASTRUCTURE ONLY.  A plain-C, one-environment-at-a-time restatement of the reference's
 * Python step (vmoens/mujoco-torch, mujoco_torch/_src/forward.py:463-496 and the modules it calls).
 * Used by tests/ as the checker for the HIP kernels, by __graft_entry__.smoke(), and by bench.py
 * as the "port" cpu_baseline.  The product package never imports, links or calls it.
 *
 * Pinning: oracle/gen_golden.py runs the reference's own Python here in the build container
 * (through oracle/ref_harness.py) and writes tests/golden/ (npz files); tests/test_oracle_golden.py checks
 * this file against those vectors.  Parity with the MuJoCo C library / MJX is UNPINNED (no mujoco
 * or jax wheel offline): the model constants come from this repo's MJCF-subset compiler.
 *
 * Same structs as the device ABI (include/mjhip.h), with HOST pointers.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/mjhip.h"

#define MJO_KNIFE_BAND_F64 1e-11
#define MJO_KNIFE_BAND_F32 1e-4

/* contact hint of the next run (see "narrow-phase ties" in mjoracle_impl.h): the caller's expected contact leaves
   [B, ncon(,3|,9)] in the run's dtype, and an int32 [B] that receives the number of pairs resolved non-naturally */
static const void *g_hint_dist, *g_hint_pos, *g_hint_frame;
static int32_t* g_tie_pairs;
void mjo_set_contact_hint(const void* dist, const void* pos, const void* frame, int32_t* tie_pairs) {
  g_hint_dist = dist; g_hint_pos = pos; g_hint_frame = frame; g_tie_pairs = tie_pairs;
}

/* RK4 stage ties (mjoracle_impl.h, stage_tie_flip): bit mask of the stage tie events (per environment, first 32) that take their
   second candidate in the next run (0: none), and an int32 [B] that receives the number of such events met */
static unsigned g_stage_tie_flip = 0;
static int32_t* g_stage_ties;
void mjo_set_stage_tie_flip(unsigned mask, int32_t* counts) { g_stage_tie_flip = mask; g_stage_ties = counts; }

/* work counters of the next run (diagnostics for the solver kernels' load balance): int32 [B, 4] = solver calls, solver iterations,
   line-search iterations, non-trivial constraint rows summed over the calls; NULL = off */
static int32_t* g_work_stats;
void mjo_set_work_stats(int32_t* stats) { g_work_stats = stats; }

/* the line search's knife band (mjoracle_impl.h, linesearch): a candidate whose derivative is below band x |d0(0)| counts as a rounding-noise candidate.
   Defaults justified by the histogram below (profiles/r06/knife_hist_*.txt); settable for the experiment "no verdict changes at another band" */
static double g_knife_band_f64 = MJO_KNIFE_BAND_F64, g_knife_band_f32 = MJO_KNIFE_BAND_F32;
void mjo_set_knife_band(double f64, double f32) { g_knife_band_f64 = f64 > 0 ? f64 : MJO_KNIFE_BAND_F64; g_knife_band_f32 = f32 > 0 ? f32 : MJO_KNIFE_BAND_F32; }
void mjo_get_knife_band(double* f64, double* f32) { *f64 = g_knife_band_f64; *f32 = g_knife_band_f32; }
/* histogram of |d0| / |d0(0)| over EVERY line-search candidate of the next runs (tools/knife_hist.py): int64 [MJO_KNIFE_BINS] -- bin 0: exactly zero,
   bin 1 + k: 1e(k - 30) <= ratio < 1e(k - 29) (k = 0 collects everything below 1e-29), last bin: ratio >= 1e2; a second array of the same shape counts
   only candidates that are NOT an end point of the current bracket (the ones the band can flag); NULL = off */
#define MJO_KNIFE_BINS 34
static int64_t *g_knife_hist, *g_knife_hist_fresh;
void mjo_set_knife_hist(int64_t* all, int64_t* fresh) { g_knife_hist = all; g_knife_hist_fresh = fresh; }
static void mjo_knife_hist_add(double d0, double d00, int fresh) {
  if (!g_knife_hist) return;
  double r = fabs(d0) / (fabs(d00) + 1e-300);
  int bin;
  if (d0 == 0) bin = 0;
  else { int k = (int)floor(log10(r)) + 30; if (k < 0) k = 0; if (k > 32) k = 32; bin = 1 + k; }
#pragma omp atomic
  g_knife_hist[bin]++;
  if (fresh && g_knife_hist_fresh) {
#pragma omp atomic
    g_knife_hist_fresh[bin]++;
  }
}

#define REAL double
#define SFX _f64
#include "mjoracle_impl.h"
#include "mjoracle_driver.h"
#undef REAL
#undef SFX
#undef R_SQRT
#undef R_SIN
#undef R_COS
#undef R_ATAN2
#undef R_POW
#undef R_FABS
#undef R_EXP

#define REAL float
#define SFX _f32
#define REAL_IS_FLOAT 1
#include "mjoracle_impl.h"
#include "mjoracle_driver.h"
#undef REAL
#undef SFX

/* knife (optional, [B] int32): per-env count of rounding-noise line-search candidates; knife_policy: see mjoracle_impl.h */
int mjo_forward(const mjhModelDesc* m, const mjhData* in, mjhData* out, int64_t B, int dtype, int stages, int flags, int nthreads, int32_t* knife, int knife_policy) {
  if (m->abi_version != MJH_ABI_VERSION) return -22;
  return dtype == MJH_F64 ? mjo_run_f64(m, in, out, B, stages, flags, 0, nthreads, knife, knife_policy) : mjo_run_f32(m, in, out, B, stages, flags, 0, nthreads, knife, knife_policy);
}

int mjo_step(const mjhModelDesc* m, const mjhData* in, mjhData* out, int64_t B, int dtype, int flags, int nthreads, int32_t* knife, int knife_policy) {
  if (m->abi_version != MJH_ABI_VERSION) return -22;
  return dtype == MJH_F64 ? mjo_run_f64(m, in, out, B, MJH_STAGE_ALL, flags, 1, nthreads, knife, knife_policy) : mjo_run_f32(m, in, out, B, MJH_STAGE_ALL, flags, 1, nthreads, knife, knife_policy);
}

int mjo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
