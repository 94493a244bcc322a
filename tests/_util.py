"""Shared helpers: load golden cases (tests/golden/*.npz) and compare Data leaves."""
import json
import os

import numpy as np
import torch

import mujoco_torch_amd as mt
from mujoco_torch_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

REAL_LEAVES = native.LISTS["MJH_DATA_REALS"]
INT_LEAVES = native.LISTS["MJH_DATA_I32"] + native.LISTS["MJH_DATA_I64"]
EXTRA_INPUTS = ["sensordata", "cacc", "cfrc_int", "subtree_linvel", "subtree_angmom"]  # inputs only some recordings set (oracle/gen_golden.py)
INPUT_LEAVES = ["time", "qpos", "qvel", "act", "qacc_warmstart", "ctrl", "qfrc_applied", "xfrc_applied", "qacc", "subtree_com", "mocap_pos", "mocap_quat"]

OUTLIER_CASES = sorted(f[:-4] for f in os.listdir(os.path.join(GOLD, "outliers")) if f.endswith(".npz")) if os.path.isdir(os.path.join(GOLD, "outliers")) else []
GOLDEN_CASES = sorted(f[:-4] for f in os.listdir(GOLD) if f.endswith(".npz") and not f.startswith(("traj_", "env_")))
ENV_CASES = sorted(f[4:-4] for f in os.listdir(GOLD) if f.endswith(".npz") and f.startswith("env_"))
TRAJECTORY_CASES = sorted(f[5:-4] for f in os.listdir(GOLD) if f.endswith(".npz") and f.startswith("traj_"))


# solver-output tolerance of goldens whose solve is ill-conditioned enough to amplify summation-order rounding beyond the default:
# CG (100 iterations) on the stiff always-active equality rows of a closed loop -- the pre-solver leaves of these cases agree
# at the default tolerance
CASE_TOL_SOL = {"equality_loops_cg_f64": 1e-5}


_MODEL_RECIPE = {}  # tables uid -> (xml, overrides, keep_sensors) of the models load_model built: the float64 twin of a float32 model


def strip_sensors(lite):
    """Drops the model's sensors (the reference cannot run rangefinders in float32: ray.py:317 keeps float64 sizes and the
    mixed-dtype dot products raise, so float32 goldens of models with rangefinders were recorded without sensors)."""
    lite.nsensor = 0
    lite.nsensordata = 0
    for k in ("sensor_type", "sensor_dim", "sensor_adr", "sensor_objid", "sensor_objtype", "sensor_needstage", "sensor_datatype", "sensor_reftype", "sensor_refid"):
        setattr(lite, k, np.zeros(0, dtype=np.int32))
    lite.sensor_cutoff = np.zeros(0)
    return lite


def load_model(xml, overrides=None, dtype=torch.float64, keep_sensors=True):
    lite = mt.mjcf.from_xml_path(mt.test_data_path(xml + ".xml"))
    for k, v in (overrides or {}).items():
        if k.startswith("model."):  # an edit of the compiled model itself (integer arrays keep their dtype): {"model.wrap_type": [...]}
            cur = getattr(lite, k[6:])
            setattr(lite, k[6:], np.array(v, dtype=np.asarray(cur).dtype) if isinstance(v, list) else v)
            continue
        setattr(lite.opt, k, np.array(v, dtype=np.float64) if isinstance(v, list) else v)
    if not keep_sensors:
        strip_sensors(lite)
    mx = mt.device_put(lite, dtype=None if dtype == torch.float64 else dtype)
    _MODEL_RECIPE[mx.tables.uid] = (xml, dict(overrides or {}), keep_sensors)
    return mx


class Golden:
    def __init__(self, case):
        self.z = np.load(os.path.join(GOLD, case + ".npz"))
        self.meta = json.loads(str(self.z["meta"]))
        self.dtype = getattr(torch, self.meta["dtype"])
        self.model = load_model(self.meta["xml"], self.meta["overrides"], self.dtype, keep_sensors=self.meta.get("keep_sensors", False))
        self.nenv, self.nsteps = self.meta["nenv"], self.meta["nsteps"]
        self.fixed_iterations = bool(self.meta.get("fixed_iterations", False))  # step(..., fixed_iterations=True) recordings

    def input_data(self, env=None):
        """Data for one env, or all envs stacked on a leading batch dim (env=None)."""
        def one(e):
            d = mt.make_data(self.model)
            if self.dtype != torch.float64:
                d = d.to(self.dtype)
            kw = {n: torch.from_numpy(self.z[f"in/{e}/{n}"].copy()) for n in INPUT_LEAVES}
            kw.update({n: torch.from_numpy(self.z[f"in/{e}/{n}"].copy()) for n in EXTRA_INPUTS if f"in/{e}/{n}" in self.z.files})
            return d.replace(**kw)

        if env is not None:
            return one(env)
        return torch.stack([one(e) for e in range(self.nenv)])

    def expected(self, env, step, name):
        return self.z[f"out/{env}/{step}/{name}"]  # every ABI leaf is recorded (oracle/gen_golden.py writes the full list)


def leaf(d, name):
    return native.data_field_tensor(d, name)


SOLVER_FLOOR = 1e-3  # solver outputs (accelerations, forces) are O(1..1e3); below 1e-3 they are solver-tolerance noise


def solver_floor(name, want_env):
    """Scale floor of one solver leaf of ONE environment.  qfrc_constraint = J^T efc_force is a sum of force terms that can cancel
    (the ant's four always-penetrating leg pairs carry 3.5e5 each and cancel to 1e-11 of rounding residue): its error is read on the
    scale of the forces it is made of, not on the scale of what is left of them."""
    if name == "qfrc_constraint":
        f = np.asarray(want_env["efc_force"], dtype=np.float64)
        return max(SOLVER_FLOOR, float(np.abs(f).max()) if f.size else 0.0)
    return SOLVER_FLOOR


def solver_err(got_env, want_env, names=None):
    """Worst per-leaf error of the solver-dependent leaves of one environment (got_env / want_env: {leaf: array of that environment})."""
    return max(rel_err(got_env[n], want_env[n], solver_floor(n, want_env)) for n in (names or SOLVER_LEAVES))


def rel_err(got, want, floor=1e-6):
    if np.asarray(want).dtype == np.float32:
        floor = max(floor, 1e-3)  # float32 leaves that are all cancellation residue (|x| ~ 1e-8 from O(1) terms) carry no relative information
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    want = np.asarray(want, dtype=np.float64).reshape(-1)
    if want.size == 0:
        return 0.0
    assert got.shape == want.shape, (got.shape, want.shape)
    scale = max(float(np.abs(want).max()), floor)  # absolute floor: all-zero leaves carry 1e-17 cancellation noise
    return float(np.abs(got - want).max() / scale)


def assert_leaves_close(get_got, get_want, tol, names=REAL_LEAVES, what=""):
    bad = []
    for n in names:
        e = rel_err(get_got(n), get_want(n))
        if not (e <= tol):
            bad.append((n, e))
    assert not bad, f"{what}: leaves beyond tol {tol:g}: {bad[:8]}"


def assert_ints_equal(get_got, get_want, what=""):
    for n in INT_LEAVES:
        g, w = np.asarray(get_got(n)), np.asarray(get_want(n))
        assert g.shape == w.shape and np.array_equal(g, w), f"{what}: integer leaf {n} differs"

SOLVER_LEAVES = ["qacc", "qacc_warmstart", "efc_force", "qfrc_constraint", "qpos", "qvel", "act", "time"]
MAX_KNIFE_POLICIES = 14
DEEP_KNIFE_POLICIES = 64
ULP_JITTER_RUNS = 16
MAX_STAGE_TIE_PAIRS = 16


HINT_LEAVES = ("contact_dist", "contact_pos", "contact_frame")


def oracle_alternatives(model, d, step=True, hint=None, knife_out=None, max_policies=None, **kw):
    """Oracle outputs under every admissible rounding outcome of the line search's noise candidates.

    The reference accepts a line-search candidate whose derivative is +-1e-13 but rejects one whose
    derivative rounds to exactly 0.0 (solver.py:440-449); which of the two happens is decided by the
    summation order of the implementation.  Returns [natural, policy 0, policy 1, ...].

    ``hint``: the outputs under test ({leaf: array}); forwarded as the oracle's contact hint so that index
    selections of the convex narrow phase that rounding noise decides (argmax over exactly symmetric or
    degenerate candidates, collision_convex.py:218-234, :532, :589) resolve to the admissible outcome the
    outputs show (pyoracle.run)."""
    import pyoracle

    if hint is not None and model.constraint_sizes_py[3] > 0:  # convex pairs: index ties; sphere / capsule pairs: coincident centres
        kw["contact_hint"] = {k: hint[k] for k in HINT_LEAVES}

    B = int(np.prod(d.qpos.shape[:-1])) if d.qpos.ndim > 1 else 1
    knife = np.zeros(B, dtype=np.int32)
    outs = [pyoracle.run(model, d, step=step, knife=knife, **kw)]
    if knife_out is not None:
        knife_out[:] = knife  # noise candidates met on the natural run, per environment
    for pol in range(max_policies or MAX_KNIFE_POLICIES):
        outs.append(pyoracle.run(model, d, step=step, knife=knife, knife_policy=pol, **kw))
        if int(knife.max()) <= pol:  # fewer noise candidates than the policy index: every one was rejected
            break
    if step and int(model.opt.integrator) == 1 and any(p[0] >= 5 for p in model.tables.pairs):
        # narrow-phase ties inside RK4 stages 1..3: those contacts are never returned, so no hint reaches them -- enumerate the
        # single flips of the (few) tie events of a step, and the double flips when there are not many
        ties = np.zeros(B, dtype=np.int32)
        pyoracle.run(model, d, step=step, stage_ties=ties, **kw)
        n = min(int(ties.max()), 32)
        masks = [1 << i for i in range(n)]
        if n <= MAX_STAGE_TIE_PAIRS:
            masks += [(1 << i) | (1 << j) for i in range(n) for j in range(i + 1, n)]
        for mk in masks:
            outs.append(pyoracle.run(model, d, step=step, stage_tie_flip=mk, **kw))
    return outs


def _model_as(model, dtype):
    xml, ov, keep = _MODEL_RECIPE[model.tables.uid]
    return load_model(xml, ov, dtype, keep_sensors=keep)


def ulp_jitter_runs(model, d_env, got_env=None, n=None, seed=0):
    """The oracle on `n` copies of ONE environment's inputs (d_env: batch of 1), every floating-point input entry moved by -1 / 0 / +1 ulp (seeded), and on the inputs
    as they are: ({leaf: array[n, ...]}, {leaf: array[1, ...]}).  How stable the reference's own outcome is at this state under last-bit changes of what it is given."""
    import pyoracle

    n = n or ULP_JITTER_RUNS
    rng = np.random.RandomState(seed)
    u = 2.0 ** -23 if d_env.qpos.dtype == torch.float32 else 2.0 ** -52
    dn = torch.cat([d_env] * n)

    def jig(x):
        if not isinstance(x, torch.Tensor) or not x.is_floating_point() or x.numel() == 0:
            return x
        return x * (1 + torch.tensor(rng.randint(-1, 2, size=tuple(x.shape)), dtype=x.dtype) * u)

    dn = dn.replace(**{k: jig(getattr(dn, k)) for k in ("qpos", "qvel", "qacc_warmstart", "ctrl", "qfrc_applied", "xfrc_applied", "act")})
    kw1, kwn = {}, {}
    if got_env is not None and model.constraint_sizes_py[3] > 0:
        kw1["contact_hint"] = {k: got_env[k] for k in HINT_LEAVES}
        kwn["contact_hint"] = {k: np.concatenate([got_env[k]] * n) for k in HINT_LEAVES}
    runs = pyoracle.run(model, dn, step=True, **kwn)
    if kwn:
        # ... and once more without the hint: it pins the normal of a sphere / capsule pair with coincident centres to the one the outputs show, which is the very quantity
        # a last-bit change moves (the ant's always-penetrating pairs: |p2 - p1| = 6e-9); the runs of both kinds are returned together (2 n)
        free = pyoracle.run(model, dn, step=True)
        runs = {k: np.concatenate([np.asarray(runs[k]), np.asarray(free[k])]) for k in runs}
    return runs, pyoracle.run(model, d_env, step=True, **kw1)


def gpu_out_to_numpy(d):
    """Data on the GPU -> {abi leaf: numpy} (host)."""
    return {n: leaf(d, n).detach().cpu().numpy() for n in REAL_LEAVES + INT_LEAVES}


PRE_SOLVER = [n for n in REAL_LEAVES if n not in SOLVER_LEAVES]


STATE_ELEMENTWISE = ("qpos", "qvel", "qacc")


def elementwise_err(got, want):
    """Worst element of max_i |got_i - want_i| / max(|want_i|, 1e-3 * |want|max): the max-norm metric (rel_err) checks an entry a thousand times smaller than
    its leaf's largest to 1e-5 of itself; this one holds every entry to the tolerance down to 1e-3 of the leaf's scale."""
    g, w = np.asarray(got, dtype=np.float64).reshape(-1), np.asarray(want, dtype=np.float64).reshape(-1)
    if w.size == 0:
        return 0.0
    scale = np.maximum(np.abs(w), 1e-3 * max(float(np.abs(w).max()), 1e-3))
    return float((np.abs(g - w) / scale).max())


def elementwise_err_per_env(got, want, batched):
    """elementwise_err with the scale of every environment's OWN slice of the leaf (round 5: the leaves upstream of the solver are held entry by entry too)."""
    g, w = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    if w.size == 0:
        return 0.0
    g, w = (g.reshape(g.shape[0], -1), w.reshape(w.shape[0], -1)) if batched else (g.reshape(1, -1), w.reshape(1, -1))
    if w.shape[1] == 0:
        return 0.0
    scale = np.maximum(np.abs(w), 1e-3 * np.maximum(np.abs(w).max(1, keepdims=True), 1e-3))
    return float((np.abs(g - w) / scale).max())


def compare_with_oracle(model, d_cpu, got, step=True, max_policies=None, **kw):
    """HIP outputs `got` (dict of numpy, batched) vs the oracle on the same inputs, without judging: per-leaf errors of the leaves
    upstream of the solver (natural oracle run), integer equality, and per environment the solver-leaf error against the
    natural oracle branch and against the closest admissible branch (oracle_alternatives)."""
    tie_pairs = None
    if model.constraint_sizes_py[3] > 0:
        B0 = int(np.prod(d_cpu.qpos.shape[:-1])) if d_cpu.qpos.ndim > 1 else 1
        tie_pairs = np.zeros(B0, dtype=np.int32)
        kw = dict(kw, tie_pairs=tie_pairs)
    knife = np.zeros(int(np.prod(d_cpu.qpos.shape[:-1])) if d_cpu.qpos.ndim > 1 else 1, dtype=np.int32)
    alts = oracle_alternatives(model, d_cpu, step=step, hint=got, knife_out=knife, max_policies=max_policies, **kw)
    nat = alts[0]
    pre = {n: rel_err(got[n], nat[n]) for n in PRE_SOLVER}
    pre_elem = {n: elementwise_err_per_env(got[n], nat[n], d_cpu.qpos.ndim > 1) for n in PRE_SOLVER}
    ints_ok = all(np.asarray(got[n]).shape == np.asarray(nat[n]).shape and np.array_equal(got[n], nat[n]) for n in INT_LEAVES)
    batched = d_cpu.qpos.ndim > 1
    B = d_cpu.qpos.shape[0] if batched else 1
    err_nat, err_best, which = np.zeros(B), np.zeros(B), np.zeros(B, dtype=np.int32)
    need = set(SOLVER_LEAVES) | {"efc_force"}
    env_of = (lambda a, e: {n: a[n][e] for n in need}) if batched else (lambda a, e: {n: a[n] for n in need})
    leaf_nat = {n: 0.0 for n in SOLVER_LEAVES}
    elem_best = np.zeros(B)  # element-wise error of the state leaves on each environment's accepted branch
    for e in range(B):
        ge = env_of(got, e)
        errs = [solver_err(ge, env_of(a, e)) for a in alts]
        err_nat[e], err_best[e], which[e] = errs[0], min(errs), int(np.argmin(errs))
        be = env_of(alts[int(which[e])], e)
        elem_best[e] = max(elementwise_err(ge[n], be[n]) for n in STATE_ELEMENTWISE)
        ne = env_of(nat, e)
        for n in SOLVER_LEAVES:
            leaf_nat[n] = max(leaf_nat[n], rel_err(ge[n], ne[n], solver_floor(n, ne)))
    return dict(pre=pre, pre_elem=pre_elem, pre_worst=max(pre.values()) if pre else 0.0, ints_ok=ints_ok, err_nat=err_nat, err_best=err_best, which=which,
                n_alts=len(alts), tie_pairs=tie_pairs, leaf_nat=leaf_nat, alts=alts, knife=knife, elem_best=elem_best)


def check_against_oracle(model, d_cpu, got, tol_pre, tol_solver, what="", step=True, max_alt_frac=1.0, max_tie_frac=1.0, band=None, tail_rules=False, tail_out=None, quantile_tol=None, **kw):
    """HIP outputs `got` (dict of numpy, batched) vs the oracle on the same inputs.

    * leaves upstream of the solver and all integer leaves: must agree outright (tol_pre / exact);
    * solver-dependent leaves, per environment: must agree with the oracle under ONE admissible rounding
      outcome of the line search's noise candidates (oracle_alternatives) within tol_solver;
    * at most `max_alt_frac` of the environments may need a non-natural branch, and at most `max_tie_frac` a non-natural
      narrow-phase tie outcome (the part of the oracle run that is steered by the outputs under test).
    * ``band`` (for steps with SEVERAL knife-edged solves -- RK4 stages, a few capped iterations -- whose 2^k outcomes cannot be
      enumerated): an environment beyond tol_solver is still accepted when the natural run met noise candidates there and its
      error is within ``band`` x the spread of the oracle's own enumerated outcomes for that environment, i.e. inside the
      reference's implementation-defined band (tests/test_oracle_golden.py pins that such bands collapse once the solve converges).
    * ``tail_rules`` (the large differential campaign only, DESIGN.md section 4 "the campaign's tail"): an environment that still matches nothing is put through the
      evidence rules below -- last-bit sensitivity of the oracle itself (`ulp`, `ulp_band`), the float64 yardstick for float32 (`f64`), equal objective value (`cost`), and the
      conditioning-aware bound on a contact normal between coincident points (`frame_cond`); ``tail_out`` receives the number of environments each rule accepted.
    Returns (fraction of environments on a non-natural line-search branch, worst solver-leaf error on the accepted branch)."""
    c = compare_with_oracle(model, d_cpu, got, step=step, **kw)
    c["tail"] = {"deep": 0, "f64": 0, "cost": 0}  # environments accepted by the rules below (reported by the campaign)
    tail_env = np.zeros(len(c["err_best"]), dtype=bool)  # ... by the rules that carry their own evidence (they do not lean on a noise candidate of the natural run)
    if d_cpu.qpos.ndim > 1 and (c["err_best"] > tol_solver).any():
        # the enumeration of a batch stops at MAX_KNIFE_POLICIES noise candidates (and sizes its stage-tie flips by the environment with the MOST tie events): an
        # environment that matched no outcome is enumerated again on its own, DEEP_KNIFE_POLICIES deep (campaign, B = 2048: a capped Newton solve with 24 noise
        # candidates whose 23rd decides the branch the GPU took)
        for e in np.nonzero(c["err_best"] > tol_solver)[0]:
            one = compare_with_oracle(model, d_cpu[int(e) : int(e) + 1], {n: got[n][int(e) : int(e) + 1] for n in got}, step=step, max_policies=DEEP_KNIFE_POLICIES,
                                      **{k: v for k, v in kw.items() if k != "tie_pairs"})
            if one["err_best"][0] < c["err_best"][e]:
                c["err_best"][e], c["elem_best"][e] = one["err_best"][0], one["elem_best"][0]  # (the element-wise figure belongs to the branch that was accepted)
                c["tail"]["deep"] += int(one["err_best"][0] <= tol_solver)
    if tail_rules and d_cpu.qpos.ndim > 1 and step:
        need = set(SOLVER_LEAVES) | {"efc_force"}
        for e in np.nonzero(c["err_best"] > tol_solver)[0]:
            e = int(e)
            de, ge = d_cpu[e : e + 1], {n: got[n][e : e + 1] for n in got}
            g_env = {n: np.asarray(ge[n])[0] for n in need}
            # (1) last-bit sensitivity of the reference itself: the oracle re-run on ULP_JITTER_RUNS copies of this environment's inputs with every floating-point
            # entry moved by -1 / 0 / +1 ulp.  Where its OWN outcome is not stable under that (a discrete decision -- an active set, a capped iteration, a narrow-phase
            # tie -- sits on a rounding edge), the outputs are accepted when they are the outcome of one such run ("ulp"), or lie inside 4 x the spread the runs show ("ulp_band")
            runs, nat = ulp_jitter_runs(model, de, ge)
            nat_env = {n: np.asarray(nat[n])[0] for n in need}
            nruns = len(np.asarray(runs["qacc"]))
            errs = [solver_err(g_env, {n: np.asarray(runs[n])[k] for n in need}) for k in range(nruns)]
            spread = max(solver_err({n: np.asarray(runs[n])[k] for n in need}, nat_env) for k in range(nruns))
            if min(errs) <= tol_solver:
                c["err_best"][e], c["err_nat"][e], c["elem_best"][e] = min(errs), np.inf, 0.0
                c["tail"]["ulp"] = c["tail"].get("ulp", 0) + 1
                tail_env[e] = True
                continue
            if spread > tol_solver and c["err_best"][e] <= 4 * spread:
                c["err_best"][e], c["err_nat"][e], c["elem_best"][e] = 0.0, np.inf, 0.0
                c["tail"]["ulp_band"] = c["tail"].get("ulp_band", 0) + 1
                tail_env[e] = True
                continue
            if d_cpu.qpos.dtype != torch.float64:
                # (2) float32: where the float32 oracle is itself further from the float64 solution of the same (upcast) inputs than the tolerance, its rounding path is
                # not the yardstick -- the outputs are held to the float64 oracle instead, at the same tolerance
                import pyoracle

                m64 = _model_as(model, torch.float64)
                o64 = pyoracle.run(m64, de.to(torch.float64), step=True)
                err64 = solver_err({n: np.asarray(ge[n], dtype=np.float64)[0] for n in need}, {n: o64[n][0] for n in need})
                if err64 <= tol_solver:
                    c["err_best"][e], c["err_nat"][e] = err64, np.inf
                    c["tail"]["f64"] += 1
                    tail_env[e] = True
            elif c["knife"][e] > 0 and c["err_best"][e] <= 100 * tol_solver and model.constraint_sizes_py[3] == 0:
                # (3) float64, a solve that met noise candidates and ended within 100 x the tolerance of the oracle's: accepted when the reference's own stopping rule
                # (improvement / scale < opt.tolerance, solver.py:501-508) cannot tell the two results apart -- same objective value to that tolerance.  (Which of two
                # bracket ends with equal costs the search returns, and with which sign a derivative at the rounding floor is accepted, are not enumerated.)
                natb = {n: np.asarray(c["alts"][0][n])[e : e + 1] for n in c["alts"][0]}
                scale = float(model.stat.meaninertia) * max(1, int(model.nv))
                gap = abs(float(solve_cost(model, dict(natb, qacc=ge["qacc"]))[0]) - float(solve_cost(model, natb)[0])) / scale
                if gap <= float(model.opt.tolerance):
                    c["err_best"][e], c["err_nat"][e], c["elem_best"][e] = 0.0, np.inf, 0.0
                    c["tail"]["cost"] += 1
                    tail_env[e] = True
    if band is not None:
        batched = d_cpu.qpos.ndim > 1
        for e in np.nonzero(c["err_best"] > tol_solver)[0]:
            need = set(SOLVER_LEAVES) | {"efc_force"}
            env_of = (lambda a: {n: a[n][e] for n in need}) if batched else (lambda a: {n: a[n] for n in need})
            spread = max(solver_err(env_of(a), env_of(c["alts"][0])) for a in c["alts"])
            assert c["knife"][e] >= 2 and c["err_best"][e] <= band * spread, f"{what} env {e}: error {c['err_best'][e]:.2e} outside {band} x the oracle's own spread {spread:.2e} (knife {c['knife'][e]})"
            c["err_best"][e] = 0.0
            c["err_nat"][e] = np.inf  # counted as off the natural branch
    bad = [(n, e) for n, e in c["pre"].items() if not (e <= tol_pre)]
    if tail_rules and [n for n, _ in bad] == ["contact_frame"] and d_cpu.qpos.ndim > 1:
        # a contact normal is (p2 - p1) / |p2 - p1|: between two spheres / capsule segments whose closest points all but coincide (the bundled ant's always-penetrating
        # leg pairs: |p2 - p1| = 6e-9 at dist = -0.16) the last bit of the two positions is 1e-8 of the normal.  Each frame is held to 64 eps x (position scale / separation)
        # there, separation = dist + r1 + r2; every other pair type and every well-separated pair keeps tol_pre.
        gt, gs = np.asarray(model.geom_type), np.asarray(model.geom_size, dtype=np.float64)
        g1, g2 = np.asarray(got["contact_geom1"]), np.asarray(got["contact_geom2"])
        nat = c["alts"][0]
        fr_g, fr_w = np.asarray(got["contact_frame"], dtype=np.float64), np.asarray(nat["contact_frame"], dtype=np.float64)
        fr_g, fr_w = fr_g.reshape(fr_g.shape[0], -1, 9), fr_w.reshape(fr_w.shape[0], -1, 9)
        err = np.abs(fr_g - fr_w).max(2)
        round_pair = np.isin(gt[g1], (2, 3)) & np.isin(gt[g2], (2, 3))
        sep = np.abs(np.asarray(nat["contact_dist"], dtype=np.float64) + gs[g1, 0] + gs[g2, 0])
        L = np.abs(np.asarray(nat["contact_pos"], dtype=np.float64)).reshape(err.shape[0], -1, 3).max(2)
        eps = 2.0 ** -52 if d_cpu.qpos.dtype == torch.float64 else 2.0 ** -23
        allow = np.where(round_pair, np.maximum(tol_pre, 64 * eps * np.maximum(L, 1e-3) / np.maximum(sep, 1e-300)), tol_pre)
        if (err <= np.minimum(allow, 1e-3)).all():
            c["tail"]["frame_cond"] = c["tail"].get("frame_cond", 0) + int((err > tol_pre).any(1).sum())
            bad = []
    assert not bad, f"{what}: leaves beyond tol {tol_pre:g}: {bad[:8]}"
    if not tail_rules and d_cpu.qpos.dtype == torch.float64:
        # ... and entry by entry (round 5; float64 -- in float32 an entry of qacc_smooth or efc_J a thousand times smaller than its leaf's largest is cancellation residue: 4e-3 / 2e-2 measured): every element of every leaf upstream of the solver within 10 x tol_pre of max(|entry|, 1e-3 of its environment's largest in that leaf).
        # Measured on the full-size config 2: 8.9e-10 (contact_dist: distances of micrometres between O(1) heights, last-bit differences of the frames); its twins of configs 3 / 5: 1.7e-13 / 1.8e-16.
        # (contact_frame of all-but-coincident closest points is ill-conditioned entry-wise as it is in the max norm: the campaign's frame_cond rule, not this test's business.)
        bad_e = [(n, e) for n, e in c["pre_elem"].items() if not (e <= 10 * tol_pre) and n != "contact_frame"]
        assert not bad_e, f"{what}: entries beyond {10 * tol_pre:g} of their own magnitude: {bad_e[:8]}"
    assert c["ints_ok"], f"{what}: integer leaves differ"
    B = len(c["err_best"])
    worst_env = int(np.argmax(c["err_best"]))
    assert c["err_best"].max() <= tol_solver, f"{what} env {worst_env}: solver outputs match no admissible oracle branch: best {c['err_best'][worst_env]:.3e}, natural {c['err_nat'][worst_env]:.3e}"
    if quantile_tol is not None:  # (q, tol): a case whose worst-environment bound states an accuracy limit keeps a TIGHT bound on its high quantile, so that a regression below the worst-case bound is seen
        q, tq = quantile_tol
        eq = float(np.quantile(c["err_best"], q))
        assert eq <= tq, f"{what}: {100 * q:g} % quantile of the solver error {eq:.3e} beyond {tq:g} (worst {c['err_best'].max():.3e})"
    if d_cpu.qpos.dtype == torch.float64 and band is None:
        # ... and ELEMENT-wise on the state leaves (qpos, qvel, qacc): every entry within the tolerance of max(|entry|, 1e-3 of its leaf's largest)
        we = int(np.argmax(c["elem_best"]))
        assert c["elem_best"].max() <= tol_solver, f"{what} env {we}: a state entry is {c['elem_best'][we]:.3e} off its own magnitude on the accepted branch (bound {tol_solver:g})"
    alt = c["err_nat"] > tol_solver
    need_alt = int(alt.sum())
    # a non-natural branch is only admissible where the reference's own result is implementation-defined: the natural oracle run
    # met a line-search candidate whose derivative is rounding noise (knife > 0) or a narrow-phase tie in that environment
    tied = c["tie_pairs"] > 0 if c["tie_pairs"] is not None else np.zeros(B, dtype=bool)
    stray = np.nonzero(alt & (c["knife"] == 0) & ~tied & ~tail_env)[0]
    assert stray.size == 0, f"{what}: envs {stray[:8].tolist()} left the natural branch without a noise candidate or tie (err {c['err_nat'][stray[:4]]})"
    assert need_alt / B <= max_alt_frac, f"{what}: {need_alt}/{B} envs needed a non-natural line-search branch (bound {max_alt_frac})"
    if c["tie_pairs"] is not None:
        tied = int((c["tie_pairs"] > 0).sum())
        assert tied / B <= max_tie_frac, f"{what}: {tied}/{B} envs needed a non-natural narrow-phase tie outcome (bound {max_tie_frac})"
    if tail_out is not None:
        for k, v in c["tail"].items():
            tail_out[k] = tail_out.get(k, 0) + v
    return need_alt / B, float(c["err_best"].max())


def solve_cost(model, out):
    """The objective the constraint solver minimises (solver.py:320-357), evaluated per environment at ``out["qacc"]`` from the leaves of
    the same forward pass: 1/2 (a - a_smooth)^T M (a - a_smooth) + sum over active rows of 1/2 D (J a - aref)^2, equality rows always
    active, the others where J a - aref < 0.  Models without frictionloss rows only.  ``out``: {leaf: array [B, ...]}."""
    ne, nf, nl, ncon, nefc = model.constraint_sizes_py
    a = np.asarray(out["qacc"], dtype=np.float64)
    B, nv = a.shape
    M = np.asarray(out["qM"], dtype=np.float64).reshape(B, nv, nv)
    da = a - np.asarray(out["qacc_smooth"], dtype=np.float64)
    cost = 0.5 * np.einsum("bi,bij,bj->b", da, M, da)
    if nefc:
        J = np.asarray(out["efc_J"], dtype=np.float64).reshape(B, nefc, nv)
        r = np.einsum("brj,bj->br", J, a) - np.asarray(out["efc_aref"], dtype=np.float64)
        D = np.asarray(out["efc_D"], dtype=np.float64)
        act = r < 0
        act[:, :ne] = True
        if nf:  # frictionloss rows (solver.py:404-416): quadratic inside |r| < f / D, linear beyond
            f = np.asarray(out["efc_frictionloss"], dtype=np.float64)[:, ne : ne + nf]
            rf = f / (D[:, ne : ne + nf] + (D[:, ne : ne + nf] == 0) * 1e-15)
            rr = r[:, ne : ne + nf]
            lin_n, lin_p = (rr <= -rf) & (f > 0), (rr >= rf) & (f > 0)
            act[:, ne : ne + nf] = ~lin_n & ~lin_p
            cost = cost + (lin_n * f * (-0.5 * rf - rr) + lin_p * f * (-0.5 * rf + rr)).sum(1)
        cost = cost + 0.5 * (D * r * r * act).sum(1)
    return cost


def load_outlier(name, extra_overrides=None):
    """tests/golden/outliers/<name>.npz: the full input Data of ONE environment that the round-1 differential campaign could not
    match to a single-policy oracle branch -> (model, unbatched Data on CPU, meta)."""
    z = np.load(os.path.join(GOLD, "outliers", name + ".npz"))
    meta = json.loads(str(z["meta"]))
    dtype = getattr(torch, meta.get("dtype", "float64"))
    mx = load_model(meta["xml"], dict(meta["overrides"], **(extra_overrides or {})), dtype)
    d = mt.make_data(mx)
    if dtype != torch.float64:
        d = d.to(dtype)
    top, con = {}, {}
    for n in REAL_LEAVES + INT_LEAVES:
        path = native.DATA_PATH[n]
        (con if len(path) == 2 else top)[path[-1]] = torch.from_numpy(z["in/" + n].copy())
    for n in ("cacc", "cfrc_int", "subtree_linvel", "subtree_angmom"):  # input-only leaves some sensors read (round-4 recordings)
        if "in/" + n in z.files:
            top[n] = torch.from_numpy(z["in/" + n].copy())
    d = d.replace(**top)
    return mx, d.replace(contact=d.contact.replace(**con)), meta


def f32_accuracy_of(mx, xml, overrides, d, got_env):
    """A float32 environment against the float64 oracle of the SAME inputs (the yardstick of test_float32_stall_case_is_float32_accuracy, for ONE environment):
    solver-leaf distances GPU-vs-float32-oracle, float32-oracle-vs-float64-oracle, GPU-vs-float64-oracle, and the worst pre-solver leaf of stage 0 against the float32
    oracle.  `d`: unbatched float32 Data on the CPU, `got_env`: {leaf: the HIP step's output for that environment}."""
    import pyoracle

    assert d.qpos.dtype == torch.float32
    mx64 = load_model(xml, overrides, torch.float64)
    d2 = torch.stack([d, d])
    w32 = pyoracle.run(mx, d2, step=True, nthreads=1)
    w64 = pyoracle.run(mx64, d2.to(torch.float64), step=True, nthreads=1)
    env = lambda a: {n: np.asarray(a[n])[0] for n in SOLVER_LEAVES}
    got = {n: np.asarray(got_env[n]) for n in SOLVER_LEAVES}
    pre = max(rel_err(got_env[n], w32[n][0]) for n in PRE_SOLVER if n in got_env)
    ints = all(np.array_equal(np.asarray(got_env[n]), w32[n][0]) for n in INT_LEAVES if n in got_env)
    return dict(gpu_vs_f32_oracle=solver_err(got, env(w32)), f32_oracle_vs_f64=solver_err(env(w32), env(w64)), gpu_vs_f64=solver_err(got, env(w64)),
                pre_solver_vs_f32_oracle=pre, ints_equal=bool(ints))


def policy_spread(model, d, **kw):
    """How far apart the oracle's OWN admissible outcomes of one step are (max over the line-search knife policies of the
    solver-leaf difference to the natural run): the size of the reference's implementation-defined band at this state."""
    import pyoracle

    knife = np.zeros(1, dtype=np.int32)
    nat = pyoracle.run(model, d, knife=knife, **kw)
    spread = 0.0
    for pol in range(MAX_KNIFE_POLICIES):
        o = pyoracle.run(model, d, knife_policy=pol, **kw)
        spread = max(spread, solver_err(o, nat))
    return spread, int(knife[0])
