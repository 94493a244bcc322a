"""The CPU oracle (oracle/mjoracle.c) against golden vectors recorded from the reference's own
Python step (oracle/gen_golden.py).  This is what pins the oracle; CPU-only."""
import os
import numpy as np
import pytest
import torch

import pyoracle
from _util import (solver_err, OUTLIER_CASES, load_outlier, policy_spread, CASE_TOL_SOL, HINT_LEAVES, SOLVER_FLOOR, GOLDEN_CASES, INT_LEAVES, REAL_LEAVES, SOLVER_LEAVES, Golden, assert_ints_equal,
                   assert_leaves_close, oracle_alternatives, rel_err)

# float64: the oracle follows the reference's operation order, differences are summation order in
# BLAS/LAPACK-backed ops; float32: same, at float32 epsilon amplified by the solver.
TOL = {torch.float64: 1e-9, torch.float32: 2e-4}


PRE_SOLVER = [n for n in REAL_LEAVES if n not in SOLVER_LEAVES]


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_oracle_matches_reference_golden(case, oracle_lib):
    """Every leaf of every recorded reference step, teacher-forced (each step starts from the
    reference's own previous output).  Leaves upstream of the solver must agree outright; the
    solver's outputs must agree with the oracle under ONE admissible rounding outcome of the line
    search's noise candidates (see _util.oracle_alternatives) -- normally the natural one."""
    g = Golden(case)
    tol = TOL[g.dtype]
    natural = total = 0
    for env in range(g.nenv):
        d = g.input_data(env)
        for s in range(g.nsteps):
            want = lambda n: g.expected(env, s, n)
            what = f"{case} env{env} step{s}"
            alts = oracle_alternatives(g.model, d, hint={n: want(n) for n in HINT_LEAVES}, fixed_iterations=g.fixed_iterations)
            assert_leaves_close(lambda n: alts[0][n], want, tol, names=PRE_SOLVER, what=what)
            assert_ints_equal(lambda n: alts[0][n], want, what=what)
            gold = {n: want(n) for n in SOLVER_LEAVES}
            errs = [solver_err(o, gold) for o in alts]
            assert min(errs) <= CASE_TOL_SOL.get(case, tol), f"{what}: solver outputs match no admissible branch, errors {errs}"
            natural += errs[0] <= tol
            total += 1
            # teacher forcing: continue from the reference's recorded state
            out = {n: np.array(want(n)) for n in REAL_LEAVES + INT_LEAVES}
            d = pyoracle.apply(d, out)
    print(f"{case}: {natural}/{total} steps matched on the natural branch")


@pytest.mark.parametrize("case", ["humanoid_cg_f64", "ant_rk4_newton_ell_f32"])
def test_oracle_batched_equals_sequential(case, oracle_lib):
    """batched call == per-env calls bit for bit (reference test/forward_test.py:142-185 analogue)."""
    g = Golden(case)
    db = g.input_data()
    outb = pyoracle.run(g.model, db, step=True, nthreads=2)
    for env in range(g.nenv):
        out1 = pyoracle.run(g.model, g.input_data(env), step=True)
        for n in REAL_LEAVES + INT_LEAVES:
            assert np.array_equal(outb[n][env], out1[n]), n


@pytest.mark.parametrize("case", [c for c in GOLDEN_CASES if c.startswith(("mesh_contact", "convex_"))])
def test_convex_tables_match_reference(case):
    """device_put's box / mesh tables (mujoco_torch_amd/convex.py) == the ones the reference's mesh.get derived
    (vertex order, polygon start vertex, face order, edge order: all of them steer index tie-breaks downstream)."""
    g = Golden(case)
    T = g.model.tables.convex
    seen = 0
    for geom, t in enumerate(T):
        if t is None:
            assert f"convex/{geom}/face" not in g.z
            continue
        seen += 1
        assert np.array_equal(g.z[f"convex/{geom}/face"], t["face"])
        assert np.array_equal(g.z[f"convex/{geom}/edge"], t["edge"])
        want_dt = g.z[f"convex/{geom}/vert"].dtype
        assert np.array_equal(g.z[f"convex/{geom}/vert"], t["vert"].astype(want_dt))
        assert np.array_equal(g.z[f"convex/{geom}/facenormal"], t["facenormal"].astype(want_dt))
        assert g.model.geom_convex_vert[geom].dtype == g.dtype
    assert seen > 0


@pytest.mark.parametrize("name", [c for c in OUTLIER_CASES if "_euler_" in c])
def test_iteration_capped_newton_states_are_implementation_defined(name, oracle_lib):
    """The Euler outliers of the round-1 campaign (convex_primitives.xml: Newton capped at 10 iterations / 6 line-search steps):
    the natural oracle run meets dozens of line-search candidates whose derivative is rounding noise, and its own admissible
    outcomes lie 1e-6 .. 1e-1 apart -- the reference's result is implementation-defined there.  Let the same solve converge
    (100 iterations, 50 line-search steps, tolerance 1e-12) and the band collapses below 1e-8."""
    mx, d, _ = load_outlier(name)
    spread, knife = policy_spread(mx, d)
    assert knife >= 10 and spread > 1e-7, (knife, spread)
    mx2, d2, _ = load_outlier(name, dict(iterations=100, ls_iterations=50, tolerance=1e-12))
    spread2, _ = policy_spread(mx2, d2)
    assert spread2 < 1e-8, spread2


@pytest.mark.parametrize("name", [c for c in OUTLIER_CASES if "_r04_" in c])
def test_campaign_tail_rules_on_recorded_outputs(name, oracle_lib):
    """The environment-steps of the round-4 campaigns (38 cases x 2048 environments x 5 steps, x 4096 x 4) that matched no outcome of the batch enumeration, with the
    outputs the HIP step produced for them (recorded on the GPU box, `got/*`): each must be accepted by the rule recorded with it -- the checker's tail rules are
    exercised here without a GPU (tests/test_gpu_parity.py::test_pinned_campaign_outliers runs the live step through the same check)."""
    import json

    from _cases import FUZZ_BAND, FUZZ_TOL_PRE
    from _util import GOLD, check_against_oracle

    mx, d, meta = load_outlier(name)
    z = np.load(os.path.join(GOLD, "outliers", name + ".npz"))
    got2 = {k[4:]: np.stack([z[k], z[k]]) for k in z.files if k.startswith("got/")}
    d2 = torch.stack([d, d])
    tol = 5e-3 if meta["dtype"] == "float32" else 1e-8
    kw = dict(what=name, band=FUZZ_BAND.get(meta["xml"]))
    tail = {}
    check_against_oracle(mx, d2, got2, FUZZ_TOL_PRE[d.qpos.dtype], tol, tail_rules=True, tail_out=tail, **kw)
    if meta["rule"] in ("branch", "band"):
        assert not any(tail.values()), tail
    else:
        assert tail.get(meta["rule"]) == 2, (meta["rule"], tail)
        if meta["rule"] != "deep" and FUZZ_BAND.get(meta["xml"]) is None:  # (a model with a campaign band is also accepted by the band once the deeper enumeration has widened it)
            with pytest.raises(AssertionError):  # ... and without the tail rules it is the mismatch the campaign reported
                check_against_oracle(mx, d2, got2, FUZZ_TOL_PRE[d.qpos.dtype], tol, **kw)


@pytest.mark.parametrize("name", [c for c in OUTLIER_CASES if "_r05_" in c])
def test_float32_accuracy_outliers_on_recorded_outputs(name, oracle_lib):
    """the ONE environment-step of the 8192 x 4 campaign (1.25 M environment-steps, profiles/r05/fuzz_parity_8192.txt) that no outcome and no tail rule of
    the checker accepts -- the ant's RK4 + Newton elliptic float32 case, a scrambled deeply penetrating pose whose objective (3.5e8) float32 resolves to +-21.  Pinned with the GPU's
    recorded outputs and held to the float64 oracle of the SAME inputs, the yardstick of test_float32_stall_case_is_float32_accuracy: everything upstream of the solver and every
    integer leaf agrees with the float32 oracle, and the GPU is NO FURTHER from the float64 solution than the float32 oracle itself is (1.8e-2 against 5.0e-2).  No rule was
    added for it: with and without `tail_rules` the checker still reports it."""
    import json

    from _cases import FUZZ_BAND, FUZZ_TOL_PRE
    from _util import GOLD, check_against_oracle, f32_accuracy_of

    mx, d, meta = load_outlier(name)
    assert meta["rule"] == "f32_accuracy" and meta["dtype"] == "float32"
    z = np.load(os.path.join(GOLD, "outliers", name + ".npz"))
    got = {k[4:]: z[k] for k in z.files if k.startswith("got/")}
    acc = f32_accuracy_of(mx, meta["xml"], meta["overrides"], d, got)
    assert acc["ints_equal"] and acc["pre_solver_vs_f32_oracle"] <= FUZZ_TOL_PRE[torch.float32], acc
    assert acc["gpu_vs_f64"] <= acc["f32_oracle_vs_f64"], acc
    for k, v in meta["measured"].items():  # the oracle in this container reproduces what was measured when the environment was pinned
        assert acc[k] == v or abs(acc[k] - v) <= 1e-6 * max(abs(v), 1e-12), (k, acc[k], v)
    got2, d2 = {n: np.stack([a, a]) for n, a in got.items()}, torch.stack([d, d])
    for tail_rules in (False, True):
        with pytest.raises(AssertionError):
            check_against_oracle(mx, d2, got2, FUZZ_TOL_PRE[torch.float32], 5e-3, what=name, band=FUZZ_BAND.get(meta["xml"]), tail_rules=tail_rules, tail_out={})


def test_knife_band_sits_in_the_empty_decades(oracle_lib):
    """the band below which a line-search candidate counts as rounding noise (the reference branches on `d0 == 0.0`, solver.py:424-467) was 1e-8 of |d0(0)|
    in float64 with nothing that showed where the flagged candidates sit.  profiles/r06/knife_hist_humanoid.txt (BASELINE config 2 at its full batch, 163,840 environment-steps):
    every candidate below 1e-8 is below 1e-14 -- the rounding cluster of a Newton step that lands on the root of a quadratic piece -- and SIX decades [1e-14, 1e-8) are empty
    (this seeded batch, with controls: a handful of cluster candidates reach 1e-13).  The band is now 1e-11 = 100 x the cluster's upper edge.  Held here on the seeded humanoid
    batch (one-iteration CG, the knife-edged benchmark case): the decades between the cluster and the old band stay empty, and the per-environment flags are THE SAME under 1e-8,
    1e-11 and 1e-12 -- no verdict of a test depends on the choice."""
    from _cases import seeded_batch

    assert pyoracle.knife_band() == (1e-11, 1e-4)
    mx, d = seeded_batch("humanoid", {"solver": 1}, torch.float64, 256)
    hist, fresh, knife = pyoracle.knife_histogram(mx, d, 12, nthreads=4)
    lo = 1 + (-13 + 30)  # bin of [1e-13, 1e-12)
    hi = 1 + (-8 + 30)   # bin of [1e-8, 1e-7)
    assert fresh[1:lo].sum() + fresh[0] > 1000, fresh          # the cluster (and the exact zeros) is populated: the case is knife-edged
    assert fresh[lo:hi].sum() == 0, fresh[lo:hi]               # ... and nothing sits between it and the old band
    assert (knife > 0).mean() > 0.2
    for band in (1e-8, 1e-12):
        _, _, k2 = pyoracle.knife_histogram(mx, d, 12, nthreads=4, band=(band, 1e-4))
        assert np.array_equal(k2, knife), band
    assert pyoracle.knife_band() == (1e-11, 1e-4)              # (the per-run override is restored)
