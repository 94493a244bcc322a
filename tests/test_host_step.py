"""Host logic of ``step`` / ``forward`` (mujoco_torch_amd/forward.py, native.get_native_model) on CPU.

The device library is replaced by tests/_hostsim.py (same pointer structs, answered by the CPU oracle), so everything the
Python side does per call -- output slab + lazily carved leaves, the per-container pointer tables, size / dtype / layout
validation, detection of Model edits after device_put -- runs for real.  The kernels themselves are covered by `-m gpu`.
"""
import numpy as np
import pytest
import torch

import _hostsim
import mujoco_torch_amd as mt
import pyoracle
from _util import INT_LEAVES, REAL_LEAVES, leaf, load_model
from mujoco_torch_amd import DisableBit


@pytest.fixture
def sim(monkeypatch, oracle_lib):
    return _hostsim.install(monkeypatch)


def seeded(mx, B, seed=0, dtype=torch.float64):
    rng = np.random.RandomState(seed)
    d = mt.make_data(mx).expand(B).clone() if B else mt.make_data(mx)
    shape = (B,) if B else ()
    d = d.replace(qvel=torch.tensor(0.1 * rng.randn(*shape, mx.nv)), ctrl=torch.tensor(0.3 * rng.randn(*shape, mx.nu)))
    return d.to(dtype) if dtype != torch.float64 else d


def assert_same(got, want, names=REAL_LEAVES + INT_LEAVES):
    for n in names:
        g, w = leaf(got, n).numpy(), np.asarray(want[n])
        assert g.shape == w.shape and np.array_equal(g, w, equal_nan=True), n


@pytest.mark.parametrize("xml,overrides,dtype,B", [
    ("humanoid", {"solver": 1}, torch.float64, 5), ("ant", {"integrator": 1, "solver": 2, "cone": 1}, torch.float32, 3),
    ("mesh_contact", {}, torch.float64, 2), ("cartpole", {}, torch.float64, 0), ("pendula", {}, torch.float64, 2), ("sensor_rig2", {}, torch.float64, 3),
])
def test_step_and_forward_fill_every_leaf_like_the_backend(sim, xml, overrides, dtype, B):
    """The Data returned by step / forward holds, leaf for leaf, what the backend wrote (written leaves) or the caller's
    tensors (everything else), in the schema order of make_data."""
    mx = load_model(xml, overrides, dtype)
    d = seeded(mx, B, dtype=dtype)
    before = {n: leaf(d, n).clone() for n in REAL_LEAVES + INT_LEAVES}
    got = mt.step(mx, d)
    assert_same(got, pyoracle.run(mx, d, step=True))
    for n in REAL_LEAVES + INT_LEAVES:  # the caller's Data is never mutated (forward.py:473-475)
        assert torch.equal(leaf(d, n), before[n]), n
    assert list(k for k, _ in got.items()) == list(k for k, _ in d.items())
    assert list(k for k, _ in got.contact.items()) == list(k for k, _ in d.contact.items())
    assert tuple(got.batch_size) == tuple(d.batch_size) and got.qpos.shape == d.qpos.shape and got.contact.frame.shape == d.contact.frame.shape
    assert int(got.ncon) == mx.constraint_sizes_py[3] and int(got.nefc) == mx.constraint_sizes_py[4]
    fwd = mt.forward(mx, d)
    assert_same(fwd, pyoracle.run(mx, d, step=False))
    assert fwd.qvel.data_ptr() == d.qvel.data_ptr() and fwd.time.data_ptr() == d.time.data_ptr()  # forward does not integrate


def test_ownership_and_aliasing_rules(sim):
    mx = load_model("humanoid", {"solver": 1})
    d = seeded(mx, 4)
    out = mt.step(mx, d)
    assert out.xfrc_applied.data_ptr() == d.xfrc_applied.data_ptr() and out.ctrl.data_ptr() == d.ctrl.data_ptr()  # untouched: alias the input
    assert out.qpos.data_ptr() != d.qpos.data_ptr() and out.contact.dist.data_ptr() != d.contact.dist.data_ptr()      # written: fresh storage
    assert out.qacc.data_ptr() != out.qacc_warmstart.data_ptr() and torch.equal(out.qacc, out.qacc_warmstart)         # solver.py:541-548
    again = mt.step(mx, d)
    assert again.qpos.data_ptr() != out.qpos.data_ptr() and torch.equal(again.qpos, out.qpos)                         # fresh per call
    for n in ("qpos", "efc_J", "contact_frame", "contact_geom", "contact_dim"):
        t = leaf(out, n)
        assert t.is_contiguous() and t.data_ptr() % 64 == 0, n  # offsets are 256-byte multiples; the CPU allocator aligns the slab to 64


def test_rollout_with_replace_update_and_assignment_tracks_the_oracle(sim):
    """`d = step(mx, d)` in a loop, with controls swapped the three ways callers do it; each step equals the backend run on the
    same inputs, i.e. the pointer table followed every leaf replacement and no leaf went stale."""
    mx = load_model("hopper")
    B = 3
    rng = np.random.RandomState(1)
    d = seeded(mx, B)
    ref = d
    for s in range(9):
        ctrl = torch.tensor(rng.uniform(-1, 1, (B, mx.nu)))
        if s % 3 == 0:
            d = d.replace(ctrl=ctrl)
        elif s % 3 == 1:
            d.update_(ctrl=ctrl)
        else:
            d.ctrl = ctrl
        ref = ref.replace(ctrl=ctrl.clone())
        d = mt.step(mx, d)
        ref = pyoracle.apply(ref, pyoracle.run(mx, ref, step=True))
        assert torch.equal(d.qpos, ref.qpos) and torch.equal(d.qvel, ref.qvel) and torch.equal(d.ctrl, ref.ctrl), s
    d.qvel.mul_(0.5)                                   # in-place write into a (lazily carved) output leaf: same storage, seen by the next call
    ref = ref.replace(qvel=ref.qvel * 0.5)
    assert torch.equal(mt.step(mx, d).qpos, pyoracle.apply(ref, pyoracle.run(mx, ref, step=True)).qpos)
    d2 = d.replace(contact=d.contact.replace(dist=torch.full_like(d.contact.dist, 7.0)))  # replaced contact leaf: an input nobody reads, but a new pointer
    assert torch.equal(mt.step(mx, d2).qpos, mt.step(mx, d).qpos)


def test_result_is_a_full_container(sim):
    """Leaves of a result are carved lazily from one slab; every container operation must still see all of them."""
    mx = load_model("ant", {}, torch.float64)
    d = seeded(mx, 4)
    out = mt.step(mx, d)
    want = pyoracle.run(mx, d, step=True)
    assert_same(out.clone(), want)
    assert_same(out[1:3], {n: want[n][1:3] for n in want})
    assert_same(torch.stack([out[0], out[1]]), {n: want[n][:2] for n in want})
    assert out.to(torch.float32).qpos.dtype == torch.float32 and out.to(torch.float32).contact.geom1.dtype == torch.int64
    flat, spec = torch.utils._pytree.tree_flatten(out)
    back = torch.utils._pytree.tree_unflatten(flat, spec)
    assert_same(back, want)
    out2 = mt.step(mx, out)                                 # result feeds the next call straight from the slab offsets
    assert_same(out2, pyoracle.run(mx, pyoracle.apply(d, want), step=True))
    sliced = mt.step(mx, out[:2].clone())
    assert torch.equal(sliced.qpos, out2.qpos[:2])


def test_out_buffers(sim):
    mx = load_model("humanoid", {"solver": 1})
    d = seeded(mx, 4)
    a, b = d.clone(), d.clone()
    r = mt.step(mx, a, out=b)
    assert r is b
    want = pyoracle.run(mx, d, step=True)
    from mujoco_torch_amd.forward import _written_names

    assert_same(b, want, names=_written_names(mx, step=True))
    assert torch.equal(b.ctrl, d.ctrl)
    for _ in range(3):                                      # ping-pong
        mt.step(mx, b, out=a)
        a, b = b, a
    ref = d
    for _ in range(4):
        ref = pyoracle.apply(ref, pyoracle.run(mx, ref, step=True))
    assert torch.equal(b.qpos, ref.qpos)
    with pytest.raises(ValueError, match="out being the input"):
        mt.step(mx, a, out=a)
    shared = b.replace(qvel=a.qvel)
    with pytest.raises(ValueError, match="shares storage"):
        mt.step(mx, a, out=shared)
    with pytest.raises(ValueError, match="not contiguous"):
        mt.step(mx, a, out=mt.make_data(mx).expand(4))     # stride-0 destination: would be filled in a temporary
    with pytest.raises(ValueError, match="holds"):
        mt.step(mx, a, out=d[:2].clone())                   # smaller batch: the kernels would write out of bounds


def test_leaf_sizes_are_validated_before_pointers_are_handed_over(sim):
    mx = load_model("humanoid", {"solver": 1})
    d = seeded(mx, 4)
    with pytest.raises(ValueError, match="ctrl holds"):
        mt.step(mx, d.replace(ctrl=torch.zeros(mx.nu, dtype=torch.float64)))           # unbatched leaf in a batched Data
    with pytest.raises(ValueError, match="xfrc_applied holds"):
        mt.step(mx, d.replace(xfrc_applied=d.xfrc_applied[:2]))                        # short batch
    with pytest.raises(RuntimeError, match="dtype"):
        mt.step(mx, d.replace(qvel=d.qvel.float()))
    with pytest.raises(RuntimeError, match="dtype"):
        mt.step(mx, d.replace(contact=d.contact.replace(geom1=d.contact.geom1.int())))
    other = load_model("ant", {}, torch.float64)
    with pytest.raises(ValueError, match="holds"):
        mt.step(other, d)                                                               # Data of another model
    ok = mt.step(mx, d)                                                                 # and a valid call still goes through afterwards
    assert torch.isfinite(ok.qpos).all()
    bad = d.replace(ctrl=torch.zeros(mx.nu, dtype=torch.float64))
    with pytest.raises(ValueError):
        mt.step(mx, bad)
    fixed = bad.replace(ctrl=d.ctrl)
    assert torch.equal(mt.step(mx, fixed).qpos, ok.qpos)                                # the table inherited from `bad` re-reads the replaced leaf


def test_strided_inputs_are_copied_each_call(sim):
    mx = load_model("hopper")
    B = 3
    d = seeded(mx, B)
    strided = d.replace(qvel=d.qvel.t().contiguous().t(), ctrl=torch.zeros(mx.nu, B, dtype=torch.float64).t())
    assert not strided.qvel.is_contiguous()
    assert torch.equal(mt.step(mx, strided).qvel, mt.step(mx, d.replace(ctrl=torch.zeros(B, mx.nu, dtype=torch.float64))).qvel)
    strided.qvel.mul_(2.0)                                  # the strided source changes in place: the next call must see it
    want = mt.step(mx, d.replace(qvel=d.qvel * 2.0, ctrl=torch.zeros(B, mx.nu, dtype=torch.float64)))
    assert torch.equal(mt.step(mx, strided).qvel, want.qvel)
    bcast = mt.make_data(mx).expand(B)                      # stride-0 batch
    assert torch.equal(mt.step(mx, bcast).qpos, mt.step(mx, bcast.clone()).qpos)


def test_model_edits_after_device_put_reach_the_backend(sim):
    """the blob was cached per (device, dtype) on the shared tables, so `mx.replace(...)` /
    `mx.tree_replace(...)` (reference test/smooth_test.py:204) silently kept stepping the first model."""
    mx = load_model("hopper")
    d = seeded(mx, 2)
    base = mt.step(mx, d)
    built0 = sim.built

    def check(m2, differs=True):
        got = mt.step(m2, d)
        assert_same(got, pyoracle.run(m2, d, step=True))
        assert (not torch.equal(got.qpos, base.qpos)) == differs
        return got

    check(mx.replace(body_mass=mx.body_mass * 2.0))
    check(mx.tree_replace({"opt.timestep": mx.opt.timestep * 0.5}))
    check(mx.tree_replace({"opt.disableflags": mx.opt.disableflags | DisableBit.GRAVITY}))
    check(mx.tree_replace({"opt.gravity": torch.tensor([0.0, 0.0, -3.0])}))
    check(mx.replace(dof_damping=mx.dof_damping + 0.5))
    check(mx.replace(geom_friction=mx.geom_friction * 0.1))
    assert sim.built == built0 + 6
    check(mx, differs=False)                                # the original still steps the original values ...
    check(mx.replace(body_mass=mx.body_mass.clone()), differs=False)   # ... and an equal-valued copy shares its blob
    assert sim.built == built0 + 6
    m3 = mx.replace(body_mass=mx.body_mass.clone())
    check(m3, differs=False)
    m3.body_mass[2] *= 3.0                                  # in-place edit of a leaf of a model that has been stepped already
    check(m3)
    m3.opt.update_(timestep=mx.opt.timestep * 2.0)          # nested container edited in place
    check(m3)
    # options that size the static tables cannot change after device_put
    for bad in ({"opt.disableflags": mx.opt.disableflags | DisableBit.CONTACT}, {"opt.cone": mt.ConeType.ELLIPTIC},
                {"opt.disableflags": mx.opt.disableflags | DisableBit.LIMIT}):
        with pytest.raises(NotImplementedError, match="device_put again"):
            mt.step(mx.tree_replace(bad), d)


def test_unbatched_and_two_batch_dims(sim):
    mx = load_model("hopper")
    E, T = 2, 3
    flat = seeded(mx, E * T)
    two = mt.make_data(mx).expand(E, T).clone().replace(qvel=flat.qvel.reshape(E, T, -1), ctrl=flat.ctrl.reshape(E, T, -1))
    got, want = mt.step(mx, two), mt.step(mx, flat)
    assert tuple(got.qpos.shape) == (E, T, mx.nq) and tuple(got.contact.dist.shape)[:2] == (E, T) and tuple(got.batch_size) == (E, T)
    assert torch.equal(got.qpos.reshape(E * T, -1), want.qpos) and torch.equal(got.efc_J.reshape(want.efc_J.shape), want.efc_J)
    one = mt.step(mx, flat[0])
    assert one.qpos.shape == (mx.nq,) and torch.equal(one.qpos, want.qpos[0])


def test_cpu_tensors_are_still_rejected_without_the_test_backend():
    mx = load_model("cartpole")
    with pytest.raises(RuntimeError, match="HIP device"):
        mt.step(mx, mt.make_data(mx))
    with pytest.raises(RuntimeError, match="HIP device"):
        mt.forward(mx, mt.make_data(mx))


def test_compile_and_vmap_wrappers_run_the_native_step(sim):
    """The reference's published modes -- `torch.compile(torch.vmap(lambda d: step(mx, d)))` (benchmarks/bench_compile.py:39-43,
    zoo/base.py compile_step=True) and plain `torch.vmap` -- wrapped around this package's step: there is nothing for a tracing
    compiler to fuse (the step is one native launch sequence), so Dynamo's graph breaks hand the call to the same batched native
    step; results equal the direct call and the returned container feeds the next call."""
    mx = load_model("hopper")
    d = seeded(mx, 4)
    want = mt.step(mx, d)
    for wrap in (torch.vmap(lambda x: mt.step(mx, x)), torch.compile(torch.vmap(lambda x: mt.step(mx, x)), fullgraph=True),
                 torch.compile(lambda x: mt.step(mx, x), fullgraph=True)):
        got = wrap(d)
        assert tuple(got.batch_size) == (4,) and torch.equal(got.qpos, want.qpos) and torch.equal(got.contact.dist, want.contact.dist)
        assert_same(got, pyoracle.run(mx, d, step=True))
        assert list(k for k, _ in got.items()) == list(k for k, _ in d.items())
        assert torch.equal(wrap(got).qpos, mt.step(mx, want).qpos)
    two = torch.vmap(torch.vmap(lambda x: mt.step(mx, x)))(torch.stack([d[:2], d[2:]]))  # nested maps are one 2 x 2 native batch
    assert torch.equal(two.qpos.reshape(4, -1), want.qpos)
    ctrl = torch.full((mx.nu,), 0.25, dtype=torch.float64)                                # a closed-over, unmapped leaf is broadcast
    got = torch.vmap(lambda x: mt.step(mx, x.replace(ctrl=ctrl)))(d)
    assert torch.equal(got.qvel, mt.step(mx, d.replace(ctrl=ctrl.expand(4, -1).clone())).qvel)


def test_kept_state_leaves_do_not_pin_the_whole_step_output(sim):
    """every leaf of a step's output is a view of an allocation shared with other leaves; the leaves a rollout keeps
    (qpos, qvel, act, time, qacc, qacc_warmstart, sensordata) come from a small allocation of their own, so a logged `d.qpos` holds a
    few hundred bytes per environment, not the ~50 KB per environment of the full output."""
    mx = load_model("humanoid", {"solver": 1})
    B = 8
    got = mt.step(mx, seeded(mx, B))
    small = got.qpos.untyped_storage().nbytes()
    bulk = got.efc_J.untyped_storage().nbytes()
    assert got.qpos.untyped_storage().data_ptr() == got.qvel.untyped_storage().data_ptr() != got.xpos.untyped_storage().data_ptr()
    assert small <= B * 8 * (mx.nq + 4 * mx.nv + 64 * 7) and bulk > 20 * small
    for n in ("qpos", "qvel", "qacc", "qacc_warmstart", "time"):
        assert getattr(got, n).untyped_storage().nbytes() == small


def test_blobs_of_replaced_models_are_released_with_their_models(sim):
    """`mx.replace(body_mass=...)` per episode must not grow the shared blob table without bound: blobs are held weakly by
    the table (strongly by the Models that stepped them and by a four-entry most-recently-used list)."""
    import gc

    mx = load_model("hopper")
    d = seeded(mx, 2)
    mt.step(mx, d)
    T = mx.tables
    for k in range(12):
        mt.step(mx.replace(body_mass=mx.body_mass * (1.0 + 0.01 * (k + 1))), d)
    gc.collect()
    assert len(T.native_recent) <= 4 and len(T.native) <= 5   # the most recent four + the original (alive through mx itself)
    built = sim.built
    mt.step(mx, d)                                             # ... which still steps without a rebuild
    assert sim.built == built


def test_in_place_edits_of_stat_are_seen(sim):
    """`stat.meaninertia` is packed into the blob; an in-place edit has to rebuild it."""
    mx = load_model("hopper")
    d = seeded(mx, 2)
    mt.step(mx, d)
    built = sim.built
    mx.stat.update_(meaninertia=torch.tensor(float(mx.stat.meaninertia) * 3.0, dtype=torch.float64))   # the reference keeps it as a tensor leaf
    got = mt.step(mx, d)
    assert sim.built == built + 1
    assert_same(got, pyoracle.run(mx, d, step=True))
    mx.stat.meaninertia.mul_(0.5)                                                                        # ... edited in place
    got = mt.step(mx, d)
    assert sim.built == built + 2
    assert_same(got, pyoracle.run(mx, d, step=True))


def test_step_output_has_the_treespec_of_its_input(sim):
    """`_order` / ncon / nefc made the pytree context of a step's output differ from its input's."""
    from torch.utils import _pytree

    mx = load_model("hopper")
    d = seeded(mx, 2)
    got = mt.step(mx, d)
    s0, s1 = _pytree.tree_structure(d), _pytree.tree_structure(got)
    assert s0 == s1
    summed = _pytree.tree_map(lambda a, b: a + b, d, got)
    assert torch.equal(summed.qpos, d.qpos + got.qpos)


def test_models_and_data_pickle_and_copy_without_their_process_local_caches(sim):
    """ADVICE r03 (medium x 2): a stepped Model could not be pickled (its tables held a WeakValueDictionary of device blobs), and
    copy.copy / copy.deepcopy duplicated ``__dict__`` as it was -- the copy kept the original's operator key, so an edited copy was
    stepped with the ORIGINAL's values under torch.vmap / torch.compile.  The reference's tensorclass Model supports both
    (torch.save(mx), handing a Model to spawn / ParallelEnv workers)."""
    import copy
    import io
    import pickle

    from mujoco_torch_amd.types import _MODELS_BY_KEY

    mx = load_model("hopper")
    d = seeded(mx, 3)
    want = mt.step(mx, d)                                           # the caches exist: a blob, a pointer table, an operator key
    loaded = pickle.loads(pickle.dumps(mx))
    assert loaded._op_key != mx._op_key and _MODELS_BY_KEY[loaded._op_key] is loaded
    assert loaded.tables.uid != mx.tables.uid and len(loaded.tables.native) == 0 and "_native_cache" not in loaded.__dict__
    assert torch.equal(mt.step(loaded, d).qpos, want.qpos)
    buf = io.BytesIO()
    torch.save(mx, buf)
    buf.seek(0)
    assert torch.equal(mt.step(torch.load(buf, weights_only=False), d).qpos, want.qpos)
    d2 = pickle.loads(pickle.dumps(want))                             # a step's output (lazily carved leaves) travels whole
    assert_same(d2, {n: leaf(want, n).numpy() for n in REAL_LEAVES + INT_LEAVES})
    assert tuple(d2.batch_size) == (3,) and int(d2.ncon) == int(want.ncon) and list(k for k, _ in d2.items()) == list(k for k, _ in want.items())
    for cp in (copy.copy(mx), copy.deepcopy(mx)):
        assert cp._op_key != mx._op_key and _MODELS_BY_KEY[cp._op_key] is cp and "_native_cache" not in cp.__dict__
        assert torch.equal(mt.step(cp, d).qpos, want.qpos)
        cp.opt.timestep = mx.opt.timestep * 0.5                        # nested container of a shallow copy: shared with mx by design (copy.copy) ...
    assert copy.copy(mx).body_mass is mx.body_mass and copy.deepcopy(mx).body_mass is not mx.body_mass
    mx.opt.timestep = mx.opt.timestep * 2.0                            # (... so restore it)
    # an edited deep copy is stepped with ITS values through every path, the vmap / compile operator included
    heavy = copy.deepcopy(mx)
    heavy.body_mass.mul_(2.0)
    direct = mt.step(heavy, d)
    assert not torch.equal(direct.qpos, want.qpos)
    assert_same(direct, pyoracle.run(heavy, d, step=True))
    assert torch.equal(torch.vmap(lambda x: mt.step(heavy, x))(d).qpos, direct.qpos)
    assert torch.equal(torch.vmap(lambda x: mt.step(mx, x))(d).qpos, want.qpos)


def test_workspace_pool_ages_on_one_clock(sim):
    """RK4 workspaces are pooled per tables object; the LRU clock has to live with the pool -- with a clock per blob a
    new blob's fresh entries looked older than every entry of earlier blobs and were evicted first."""
    from mujoco_torch_amd import native

    workspace = _hostsim.REAL_NATIVE_MODEL.workspace                  # the product's pool logic, on the stand-in's blobs (which need no scratch themselves)
    mx = load_model("ant", {"integrator": 1})
    dev = torch.device("cpu")
    nm1 = native.get_native_model(mx, dev, torch.float64)
    nm2 = native.get_native_model(mx.replace(body_mass=mx.body_mass * 1.5), dev, torch.float64)
    assert nm2 is not nm1 and nm2._work is nm1._work
    nm1.work_bytes = nm2.work_bytes = 16
    for B in (1, 2, 3, 4):
        workspace(nm1, B)
    w5 = workspace(nm2, 5)                                            # evicts B = 1, the oldest of the pool
    assert sorted(k[0] for k in nm1._work["bufs"]) == [2, 3, 4, 5]
    workspace(nm2, 6)                                                 # ... then B = 2 -- not the entry this blob has just used
    assert sorted(k[0] for k in nm1._work["bufs"]) == [3, 4, 5, 6] and workspace(nm2, 5) is w5


def test_reassigned_unbatched_leaf_is_seen_by_the_stamp(sim):
    """``unbatched.data = new_tensor`` bumps neither a container version nor a tensor version counter."""
    from mujoco_torch_amd import native
    from mujoco_torch_amd.container import UnbatchedTensor

    mx = load_model("hopper")
    wrapped = [k for k, v in mx._fields.items() if isinstance(v, UnbatchedTensor) and isinstance(v.data, torch.Tensor)]
    s0 = native._stamp(mx)
    assert native._same_stamp(s0, native._stamp(mx))
    if wrapped:
        w = mx._fields[wrapped[0]]
        w.data = w.data.clone()
        assert not native._same_stamp(s0, native._stamp(mx))


def test_sensors_read_the_data_leaves_no_stage_writes(sim):
    """force / torque / accelerometer / subtree momentum sensors read Data.cfrc_int / cacc / subtree_linvel / subtree_angmom, which no
    stage of the reference writes (sensor.py:383-416, 261-266): the caller's values reach the library through the trailing pointers of mjhData."""
    import xml.etree.ElementTree as ET

    mx = load_model("sensor_rig2")
    assert mx.tables.sensors["extra_leaves"] == ("cacc", "cfrc_int", "subtree_angmom", "subtree_linvel")
    d = seeded(mx, 2)
    base = mt.step(mx, d)
    nb = int(mx.nbody)
    rng = np.random.RandomState(3)
    d2 = d.replace(cfrc_int=torch.tensor(rng.randn(2, nb, 6)), cacc=torch.tensor(rng.randn(2, nb, 6)), subtree_linvel=torch.tensor(rng.randn(2, nb, 3)),
                   subtree_angmom=torch.tensor(rng.randn(2, nb, 3)).transpose(0, 1).contiguous().transpose(0, 1))  # (one of them strided: copied per call)
    got = mt.step(mx, d2)
    assert_same(got, pyoracle.run(mx, d2, step=True))
    sens = list(ET.parse(mt.test_data_path("sensor_rig2.xml")).getroot().find("sensor"))
    adr = 0
    for sn in sens:
        dim = {"ballquat": 4, "framequat": 4}.get(sn.tag, 1 if sn.tag in ("touch", "jointpos", "jointvel", "tendonpos", "tendonvel", "actuatorpos", "actuatorvel", "actuatorfrc", "jointactuatorfrc",
                                                                              "tendonactuatorfrc", "jointlimitpos", "e_potential", "clock") else 3)
        changed = not torch.equal(got.sensordata[:, adr : adr + dim], base.sensordata[:, adr : adr + dim])
        assert changed == (sn.tag in ("force", "torque", "accelerometer", "subtreelinvel", "subtreeangmom")), sn.tag
        adr += dim
    assert adr == int(mx.nsensordata)
    with pytest.raises(ValueError, match="cfrc_int"):
        mt.step(mx, d.replace(cfrc_int=torch.zeros(2, nb, 5, dtype=torch.float64)))


def test_value_only_model_edits_do_not_retrace_the_compiled_step(sim):
    """the operator took the Model as a Python int, a constant of the traced graph -- every `mx.replace(body_mass=...)` (per-episode domain
    randomisation) recompiled the step until Dynamo's recompile limit.  The graph's constant is now the STRUCTURE id; the Model's values travel as a tensor input."""
    import torch._dynamo as dynamo
    import torch._dynamo.testing  # noqa: F401  (a submodule: not imported by `import torch._dynamo` alone)

    mx = load_model("hopper")
    d = seeded(mx, 3)
    holder = {"m": mx}
    dynamo.reset()
    counter = dynamo.testing.CompileCounter()
    step = torch.compile(lambda x: mt.step(holder["m"], x), fullgraph=True, backend=counter)
    base = step(d)
    assert torch.equal(base.qpos, mt.step(mx, d).qpos)
    frames = counter.frame_count
    for k in range(3):
        holder["m"] = mx.replace(body_mass=mx.body_mass * (1.5 + 0.1 * k))
        got = step(d)
        assert torch.equal(got.qpos, mt.step(holder["m"], d).qpos) and not torch.equal(got.qpos, base.qpos)
    assert counter.frame_count == frames, f"value-only Model edits retraced the step ({counter.frame_count - frames} times)"
    other = load_model("halfcheetah")                                   # another structure: its own trace, checked at run time
    holder["m"] = other
    d2 = seeded(other, 3)
    assert torch.equal(step(d2).qpos, mt.step(other, d2).qpos)


@pytest.mark.parametrize("xml,overrides", [("ant", {"integrator": 1, "solver": 2, "cone": 1}), ("sensor_rig2", {})])
def test_vmap_and_compile_carry_the_input_only_sensor_leaves(sim, xml, overrides):
    """models whose sensors read cacc / cfrc_int / subtree_linvel / subtree_angmom (ant: BASELINE config 3; the sensor rigs) raised
    under `torch.vmap(step)` / `torch.compile(vmap(step))` -- the operator rebuilt those leaves from the UNBATCHED template -- and a caller's values for them
    were dropped.  They are operator inputs now: results equal the direct call with zeros, with caller-set values, and with the leaves absent."""
    mx = load_model(xml, overrides)
    B = 4
    d = seeded(mx, B)
    nb = int(mx.nbody)
    rng = np.random.RandomState(5)
    d2 = d.replace(cacc=torch.tensor(rng.randn(B, nb, 6)), cfrc_int=torch.tensor(rng.randn(B, nb, 6)), subtree_linvel=torch.tensor(rng.randn(B, nb, 3)),
                   subtree_angmom=torch.tensor(rng.randn(B, nb, 3)))
    for x in (d, d2):
        want = mt.step(mx, x)
        assert_same(want, pyoracle.run(mx, x, step=True))
        for wrap in (torch.vmap(lambda y: mt.step(mx, y)), torch.compile(torch.vmap(lambda y: mt.step(mx, y)), fullgraph=True),
                     torch.compile(lambda y: mt.step(mx, y), fullgraph=True)):
            got = wrap(x)
            assert_same(got, {n: leaf(want, n).numpy() for n in REAL_LEAVES + INT_LEAVES})
            assert torch.equal(got.cacc, x.cacc) and torch.equal(got.subtree_angmom, x.subtree_angmom)     # untouched leaves stay the caller's
            assert torch.equal(wrap(got).sensordata, mt.step(mx, want).sensordata)
    assert not torch.equal(mt.step(mx, d2).sensordata, mt.step(mx, d).sensordata)                            # (the values matter: they were dropped before)
    cacc = torch.tensor(rng.randn(nb, 6))                                                                  # a closed-over, unmapped input-only leaf is broadcast
    got = torch.vmap(lambda y: mt.step(mx, y.replace(cacc=cacc)))(d)
    assert torch.equal(got.sensordata, mt.step(mx, d.replace(cacc=cacc.expand(B, nb, 6).clone())).sensordata)


def test_structure_registry_outlives_collected_value_only_copies(sim):
    """the structure registry held only the NEWEST Model of a structure, weakly -- when that value-only copy was collected the entry went
    with it and the next trace / recompile of a step of the ORIGINAL raised 'no Model of the structure ... exists any more'."""
    import gc

    from mujoco_torch_amd import compile_op

    mx = load_model("hopper")
    tmp = mx.replace(body_mass=mx.body_mass * 1.5)
    assert compile_op._structure(mx._struct_uid) is not None
    del tmp
    gc.collect()
    assert compile_op._structure(mx._struct_uid) is mx
    d = seeded(mx, 2)
    want = mt.step(mx, d)
    torch._dynamo.reset()
    got = torch.compile(torch.vmap(lambda x: mt.step(mx, x)), fullgraph=True)(d)   # a fresh trace: register_fake looks the structure up
    assert torch.equal(got.qpos, want.qpos)
    uid = mx._struct_uid
    del mx, got, want, d
    gc.collect()
    with pytest.raises(RuntimeError, match="no Model of the structure"):
        compile_op._structure(uid)


def test_unevaluated_sensor_slots_keep_the_callers_values(sim, tmp_path):
    """<camprojection> / <user> sensors (compiled since round 5) belong to the types sensor.py has no branch for: their sensordata slots pass through the step."""
    (tmp_path / "m.xml").write_text(
        '<mujoco><worldbody><camera name="cam" pos="0 -1 1"/><body name="b"><joint name="j" type="hinge" axis="0 1 0"/><geom size="0.1" pos="0.2 0 0"/><site name="s1" pos="0.3 0 0"/></body></worldbody>'
        '<sensor><jointpos joint="j"/><camprojection site="s1" camera="cam"/><user dim="3" needstage="vel"/><jointvel joint="j"/></sensor></mujoco>')
    mx = mt.device_put(mt.mjcf.from_xml_path(str(tmp_path / "m.xml")))
    d = seeded(mx, 3)
    sd = torch.tensor(np.random.RandomState(2).randn(3, int(mx.nsensordata)))
    d = d.replace(sensordata=sd.clone())
    got = mt.step(mx, d)
    assert_same(got, pyoracle.run(mx, d, step=True))
    assert torch.equal(got.sensordata[:, 1:6], sd[:, 1:6])                                   # camprojection (2) + user (3): untouched
    assert not torch.equal(got.sensordata[:, 0], sd[:, 0]) and not torch.equal(got.sensordata[:, 6], sd[:, 6])   # jointpos / jointvel: evaluated
