/* mjoracle_driver.h -- model conversion, workspace, per-env load/store, Euler/RK4 drivers.
 * TEST INFRASTRUCTURE (see mjoracle.c).  Included once per REAL type. */

static int FN(field_count)(const mjhModelDesc* m, const char* name) {
  int nq = m->nq, nv = m->nv, nu = m->nu, na = m->na, nb = m->nbody, nj = m->njnt, ng = m->ngeom;
  int ncon = m->ncon, nefc = m->nefc;
#define F(n, c) if (!strcmp(name, #n)) return (c);
  F(time, 1) F(qpos, nq) F(qvel, nv) F(act, na) F(qacc_warmstart, nv) F(ctrl, nu) F(qfrc_applied, nv)
  F(xfrc_applied, nb * 6) F(mocap_pos, m->nmocap * 3) F(mocap_quat, m->nmocap * 4) F(qacc, nv) F(act_dot, na)
  F(xpos, nb * 3) F(xquat, nb * 4) F(xmat, nb * 9) F(xipos, nb * 3) F(ximat, nb * 9) F(xanchor, nj * 3) F(xaxis, nj * 3)
  F(geom_xpos, ng * 3) F(geom_xmat, ng * 9) F(site_xpos, m->nsite * 3) F(site_xmat, m->nsite * 9)
  F(cam_xpos, m->ncam * 3) F(cam_xmat, m->ncam * 9) F(light_xpos, m->nlight * 3) F(light_xdir, m->nlight * 3)
  F(subtree_com, nb * 3) F(cdof, nv * 6) F(cinert, nb * 10) F(crb, nb * 10) F(ten_length, m->ntendon) F(ten_J, m->ntendon * nv) F(ten_velocity, m->ntendon) F(actuator_length, nu)
  F(actuator_moment, nu * nv) F(qM, nv * nv) F(qLD, nv * nv) F(contact_dist, ncon) F(contact_pos, ncon * 3)
  F(contact_frame, ncon * 9) F(contact_includemargin, ncon) F(contact_friction, ncon * 5) F(contact_solref, ncon * 2)
  F(contact_solreffriction, ncon * 2) F(contact_solimp, ncon * 5) F(sensordata, m->nsensordata) F(efc_J, nefc * nv) F(efc_frictionloss, nefc)
  F(efc_D, nefc) F(efc_aref, nefc) F(efc_force, nefc) F(actuator_velocity, nu) F(cvel, nb * 6) F(cdof_dot, nv * 6)
  F(qfrc_bias, nv) F(qfrc_passive, nv) F(qfrc_gravcomp, nv) F(actuator_force, nu) F(qfrc_actuator, nv) F(qfrc_smooth, nv)
  F(qacc_smooth, nv) F(qfrc_constraint, nv)
#undef F
  return 0;
}

static REAL* FN(ralloc)(size_t n) { return (REAL*)calloc(n + 8, sizeof(REAL)); }

static void FN(model_init)(FN(MjoModel) * M, const mjhModelDesc* d) {
  M->d = d;
  M->timestep = (REAL)d->timestep;
  M->impratio = (REAL)d->impratio;
  M->meaninertia = (REAL)d->meaninertia;
  M->density = (REAL)d->density; M->viscosity = (REAL)d->viscosity;
  M->wind[0] = (REAL)d->wind_x; M->wind[1] = (REAL)d->wind_y; M->wind[2] = (REAL)d->wind_z;
  M->has_fluid = (d->density > 0) || (d->viscosity > 0) || (d->wind_x != 0) || (d->wind_y != 0) || (d->wind_z != 0);
  M->has_gravcomp = 0;
  for (int b = 0; b < d->nbody; b++) if (d->body_gravcomp[b] != 0) M->has_gravcomp = 1;
  M->gravity[0] = (REAL)d->gravity_x; M->gravity[1] = (REAL)d->gravity_y; M->gravity[2] = (REAL)d->gravity_z;
#define X(n) { M->n = FN(ralloc)((size_t)d->len_##n); for (int64_t i = 0; i < d->len_##n; i++) M->n[i] = (REAL)d->n[i]; }
  MJH_MODEL_REAL_ARRAYS(X)
#undef X
}
static void FN(model_free)(FN(MjoModel) * M) {
#define X(n) free(M->n);
  MJH_MODEL_REAL_ARRAYS(X)
#undef X
}

static void FN(work_init)(FN(MjoWork) * w, const mjhModelDesc* m) {
  int nv = m->nv, nefc = m->nefc, nb = m->nbody, nq = m->nq;
  int big = nefc > nq ? nefc : nq;
  if (nv > big) big = nv;
  if (m->ntendon > big) big = m->ntendon;
#define X(n) w->n = FN(ralloc)((size_t)FN(field_count)(m, #n));
  MJH_DATA_REALS(X)
#undef X
  w->cacc = FN(ralloc)(nb * 6); w->cfrc = FN(ralloc)(nb * 6); w->sub_mass = FN(ralloc)(nb); w->sub_pos = FN(ralloc)(nb * 3);
  w->crb_cdof = FN(ralloc)(nv * 6); w->jacdiff = FN(ralloc)(6 * nv); w->tmp_nv = FN(ralloc)(nv); w->tmp_nv2 = FN(ralloc)(nv);
  w->tmp_nefc = FN(ralloc)(big); w->efc_pos = FN(ralloc)(nefc); w->efc_pos_norm = FN(ralloc)(nefc);
  w->efc_invweight = FN(ralloc)(nefc); w->efc_solref = FN(ralloc)(2 * nefc); w->efc_solimp = FN(ralloc)(5 * nefc);
  w->qM2 = FN(ralloc)(nv * nv); w->qLD2 = FN(ralloc)(nv * nv); w->H = FN(ralloc)(nv * nv); w->HL = FN(ralloc)(nv * nv);
  w->s_qacc = FN(ralloc)(nv); w->s_qfrc = FN(ralloc)(nv); w->s_Jaref = FN(ralloc)(nefc); w->s_force = FN(ralloc)(nefc);
  w->s_Ma = FN(ralloc)(nv); w->s_grad = FN(ralloc)(nv); w->s_Mgrad = FN(ralloc)(nv); w->s_search = FN(ralloc)(nv);
  w->s_mv = FN(ralloc)(nv); w->s_jv = FN(ralloc)(nefc); w->s_quad = FN(ralloc)(3 * nefc);
  w->s_prev_grad = FN(ralloc)(nv); w->s_prev_Mgrad = FN(ralloc)(nv);
  w->s_active = (unsigned char*)calloc(nefc + 8, 1);
  w->rk_qpos0 = FN(ralloc)(nq); w->rk_qvel0 = FN(ralloc)(nv); w->rk_act0 = FN(ralloc)(m->na); w->rk_qvel = FN(ralloc)(nv);
  w->rk_qacc = FN(ralloc)(nv); w->rk_actdot = FN(ralloc)(m->na); w->rk_kqvel = FN(ralloc)(nv);
  w->in_subtree_com = FN(ralloc)(nb * 3);
  w->cand_dist = FN(ralloc)(m->ncand); w->cand_pos = FN(ralloc)(3 * (size_t)m->ncand); w->cand_frame = FN(ralloc)(9 * (size_t)m->ncand);
  w->con_src = (int*)calloc((size_t)m->ncon + 8, sizeof(int));
}
static void FN(work_free)(FN(MjoWork) * w) {
#define X(n) free(w->n);
  MJH_DATA_REALS(X)
#undef X
  free(w->cacc); free(w->cfrc); free(w->sub_mass); free(w->sub_pos); free(w->crb_cdof); free(w->jacdiff); free(w->tmp_nv);
  free(w->tmp_nv2); free(w->tmp_nefc); free(w->efc_pos); free(w->efc_pos_norm); free(w->efc_invweight); free(w->efc_solref);
  free(w->efc_solimp); free(w->qM2); free(w->qLD2); free(w->H); free(w->HL); free(w->s_qacc); free(w->s_qfrc); free(w->s_Jaref);
  free(w->s_force); free(w->s_Ma); free(w->s_grad); free(w->s_Mgrad); free(w->s_search); free(w->s_mv); free(w->s_jv);
  free(w->s_quad); free(w->s_prev_grad); free(w->s_prev_Mgrad); free(w->s_active); free(w->rk_qpos0); free(w->rk_qvel0);
  free(w->rk_act0); free(w->rk_qvel); free(w->rk_qacc); free(w->rk_actdot); free(w->rk_kqvel); free(w->in_subtree_com);
  free(w->cand_dist); free(w->cand_pos); free(w->cand_frame); free(w->con_src);
}

/* snapshot of the leaves a forward() pass writes, used to keep stage-0 results across RK4 stages */
static void FN(copy_work_outputs)(const mjhModelDesc* m, FN(MjoWork) * dst, const FN(MjoWork) * src) {
#define X(n) memcpy(dst->n, src->n, sizeof(REAL) * (size_t)FN(field_count)(m, #n));
  MJH_DATA_REALS(X)
#undef X
  memcpy(dst->con_src, src->con_src, sizeof(int) * (size_t)m->ncon);
}

static void FN(step_env)(const FN(MjoModel) * M, FN(MjoWork) * w, FN(MjoWork) * w0, int flags) {
  const mjhModelDesc* m = M->d;
  int nq = m->nq, nv = m->nv, na = m->na;
  /* _check_state forward.py:44-59 */
  for (int i = 0; i < nq; i++) { REAL x = w->qpos[i]; if (!isfinite(x) || R_FABS(x) > (REAL)mjMAXVAL) w->qpos[i] = M->qpos0[i]; }
  for (int i = 0; i < nv; i++) { REAL x = w->qvel[i]; if (!isfinite(x) || R_FABS(x) > (REAL)mjMAXVAL) w->qvel[i] = 0; }
  for (int i = 0; i < nv; i++) { REAL x = w->qacc[i]; if (!isfinite(x) || R_FABS(x) > (REAL)mjMAXVAL) w->qacc[i] = 0; }
  REAL time0 = w->time[0];
  FN(forward_env)(M, w, MJH_STAGE_ALL, flags, 1);
  w->hint_dist = NULL; /* the returned contact leaves are stage 0's: later RK4 stages resolve ties naturally ... */
  w->stage_mode = 1;   /* ... or under the caller's single-flip policy (stage_tie_flip) */
  if (m->integrator == INT_EULER) { /* _euler :313-328 */
    const REAL* qacc = w->qacc;
    if (!(m->disableflags & DSBL_EULERDAMP)) {
      for (int i = 0; i < nv * nv; i++) w->qM2[i] = w->qM[i];
      for (int i = 0; i < nv; i++) w->qM2[i * nv + i] = w->qM[i * nv + i] + M->timestep * M->dof_damping[i];
      FN(cholesky)(w->qM2, w->qLD2, nv);
      for (int i = 0; i < nv; i++) w->s_grad[i] = w->qfrc_smooth[i] + w->qfrc_constraint[i];
      FN(cholesky_solve)(w->qLD2, w->s_grad, w->s_Mgrad, nv, w->tmp_nv2);
      qacc = w->s_Mgrad;
    }
    memcpy(w->rk_qpos0, w->qpos, sizeof(REAL) * nq);
    memcpy(w->rk_qvel0, w->qvel, sizeof(REAL) * nv);
    memcpy(w->rk_act0, w->act, sizeof(REAL) * na);
    memcpy(w->rk_actdot, w->act_dot, sizeof(REAL) * na);
    memcpy(w->rk_qacc, qacc, sizeof(REAL) * nv);
    FN(advance)(M, w, w->rk_qpos0, w->rk_qvel0, w->rk_act0, time0, w->rk_actdot, w->rk_qacc, NULL);
    return;
  }
  /* _rungekutta4 :331-370; w0 keeps the stage-0 Data (d_t0), w is the scratch Data stages run in */
  static const REAL A[3] = {(REAL)0.5, (REAL)0.5, (REAL)1.0};
  /* the tableau is a float32 literal tensor up-cast to the data dtype (_CachedConst, math.py:34-45) */
  const REAL Bt[4] = {(REAL)(float)(1.0 / 6.0), (REAL)(float)(1.0 / 3.0), (REAL)(float)(1.0 / 3.0), (REAL)(float)(1.0 / 6.0)};
  REAL dt = M->timestep;
  FN(copy_work_outputs)(m, w0, w);
  memcpy(w->rk_qpos0, w->qpos, sizeof(REAL) * nq);
  memcpy(w->rk_qvel0, w->qvel, sizeof(REAL) * nv);
  memcpy(w->rk_act0, w->act, sizeof(REAL) * na);
  for (int i = 0; i < nv; i++) { w->rk_kqvel[i] = w->qvel[i]; w->rk_qvel[i] = Bt[0] * w->qvel[i]; w->rk_qacc[i] = Bt[0] * w->qacc[i]; }
  for (int i = 0; i < na; i++) w->rk_actdot[i] = Bt[0] * w->act_dot[i];
  for (int s = 0; s < 3; s++) {
    REAL a = A[s], b = Bt[s + 1];
    REAL t = time0 + A[s] * dt; /* C = column sums of the (diagonal) tableau */
    for (int i = 0; i < nv; i++) w->tmp_nv2[i] = a * w->rk_kqvel[i];           /* dqvel */
    FN(integrate_pos)(M, w->rk_qpos0, w->tmp_nv2, dt, w->tmp_nefc);             /* kqpos */
    for (int i = 0; i < na; i++) w->act[i] = w->rk_act0[i] + (a * w->act_dot[i]) * dt;
    for (int i = 0; i < nv; i++) w->rk_kqvel[i] = w->rk_qvel0[i] + (a * w->qacc[i]) * dt;
    for (int i = 0; i < nq; i++) w->qpos[i] = w->tmp_nefc[i];
    for (int i = 0; i < nv; i++) w->qvel[i] = w->rk_kqvel[i];
    w->time[0] = t;
    FN(forward_env)(M, w, MJH_STAGE_ALL, flags, 0);
    for (int i = 0; i < nv; i++) { w->rk_qvel[i] = w->rk_qvel[i] + b * w->rk_kqvel[i]; w->rk_qacc[i] = w->rk_qacc[i] + b * w->qacc[i]; }
    for (int i = 0; i < na; i++) w->rk_actdot[i] = w->rk_actdot[i] + b * w->act_dot[i];
  }
  /* _advance(d_t0, act_dot, qacc, qvel): restore the stage-0 leaves then advance */
  {
    REAL *kq = w->rk_qvel, *ka = w->rk_qacc, *kd = w->rk_actdot, *q0 = w->rk_qpos0, *v0 = w->rk_qvel0, *a0 = w->rk_act0;
    FN(copy_work_outputs)(m, w, w0);
    FN(advance)(M, w, q0, v0, a0, time0, kd, ka, kq);
  }
}

static int FN(mjo_run)(const mjhModelDesc* m, const mjhData* in, mjhData* out, int64_t B, int stages, int flags, int do_step, int nthreads, int32_t* knife, int knife_policy) {
  FN(MjoModel) M;
  FN(model_init)(&M, m);
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
  nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
  {
    FN(MjoWork) w, w0;
    FN(work_init)(&w, m);
    FN(work_init)(&w0, m);
    int32_t* eq_zero = (int32_t*)calloc((size_t)m->neq + 1, sizeof(int32_t));
#pragma omp for schedule(static)
    for (int64_t e = 0; e < B; e++) {
      /* load every provided input leaf */
#define X(n) { int c = FN(field_count)(m, #n); if (in->n && c) memcpy(w.n, (const REAL*)in->n + e * c, sizeof(REAL) * c); else if (c) memset(w.n, 0, sizeof(REAL) * c); }
      MJH_DATA_REALS(X)
#undef X
      memcpy(w.in_subtree_com, w.subtree_com, sizeof(REAL) * 3 * m->nbody);
      w.knife = 0;
      w.knife_policy = knife_policy;
      w.nf = m->nf + m->nft; w.ne_nf = m->ne + m->nf + m->nft; w0.nf = w.nf; w0.ne_nf = w.ne_nf;
      w.stage_mode = 0; w.stage_tie_n = 0; w.stage_tie_flip = g_stage_tie_flip;
      w.stat_solves = w.stat_niter = w.stat_ls = w.stat_rows = 0;
      w.tie_on = 0; w.tie_n = 0; w.tie_pairs = 0; w.prim_hint_n = NULL; w.prim_adopted = 0; w0.prim_hint_n = NULL; w0.prim_adopted = 0;
      w.x_cacc = in->cacc ? (const REAL*)in->cacc + e * 6 * m->nbody : NULL;
      w.x_cfrc_int = in->cfrc_int ? (const REAL*)in->cfrc_int + e * 6 * m->nbody : NULL;
      w.x_subtree_linvel = in->subtree_linvel ? (const REAL*)in->subtree_linvel + e * 3 * m->nbody : NULL;
      w.x_subtree_angmom = in->subtree_angmom ? (const REAL*)in->subtree_angmom + e * 3 * m->nbody : NULL;
      w.eq_active = in->eq_active ? in->eq_active + e * m->neq : eq_zero;
      w0.eq_active = w.eq_active;
      w.hint_dist = g_hint_dist ? (const REAL*)g_hint_dist + e * m->ncon : NULL;
      w.hint_pos = g_hint_pos ? (const REAL*)g_hint_pos + e * m->ncon * 3 : NULL;
      w.hint_frame = g_hint_frame ? (const REAL*)g_hint_frame + e * m->ncon * 9 : NULL;
      if (do_step) FN(step_env)(&M, &w, &w0, flags);
      else FN(forward_env)(&M, &w, stages, flags, 1);
#define X(n) { int c = FN(field_count)(m, #n); if (out->n && c) memcpy((REAL*)out->n + e * c, w.n, sizeof(REAL) * c); }
      MJH_DATA_REALS(X)
#undef X
      if (knife) knife[e] = w.knife;
      if (g_tie_pairs) g_tie_pairs[e] = w.tie_pairs;
      if (g_stage_ties) g_stage_ties[e] = w.stage_tie_n;
      if (g_work_stats) { g_work_stats[4 * e] = w.stat_solves; g_work_stats[4 * e + 1] = w.stat_niter; g_work_stats[4 * e + 2] = w.stat_ls; g_work_stats[4 * e + 3] = w.stat_rows; }
      const int* cs = w.con_src; /* (RK4: restored to stage 0's with the other returned leaves) */
      if (out->contact_dim) for (int c = 0; c < m->ncon; c++) out->contact_dim[e * m->ncon + c] = m->con_dim[cs[c]];
      if (out->contact_geom1) for (int c = 0; c < m->ncon; c++) out->contact_geom1[e * m->ncon + c] = m->con_geom1[cs[c]];
      if (out->contact_geom2) for (int c = 0; c < m->ncon; c++) out->contact_geom2[e * m->ncon + c] = m->con_geom2[cs[c]];
      if (out->contact_geom) for (int c = 0; c < m->ncon; c++) { out->contact_geom[(e * m->ncon + c) * 2] = m->con_geom1[cs[c]]; out->contact_geom[(e * m->ncon + c) * 2 + 1] = m->con_geom2[cs[c]]; }
      if (out->contact_efc_address) for (int c = 0; c < m->ncon; c++) out->contact_efc_address[e * m->ncon + c] = m->con_efc_address[c];
    }
    FN(work_free)(&w);
    FN(work_free)(&w0);
    free(eq_zero);
  }
  FN(model_free)(&M);
  return 0;
}
