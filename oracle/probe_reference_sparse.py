"""Container-only probe (needs /root/reference): is the reference's SPARSE inertia path (support.is_sparse: jacobian=sparse, or
auto with nv >= 60; smooth.factor_m :321-331, solve_m :345-354, tables device.py:747-829) a parity target?

It runs the reference's own kinematics -> com_pos -> crb -> factor_m -> solve_m on bundled models forced to jacobian=sparse and
compares solve_m(y) with numpy.linalg.solve(full_m, y).  Result (profiles/r02/reference_sparse_probe.txt): 13 - 27 % error.
Cause: factor_m applies each depth group's updates with `qld.scatter(0, out, qld[out] + qld_update)` (smooth.py:325-326); `out`
repeats whenever several descendants update the same ancestor row in one group, and scatter keeps ONE of the duplicates, so the
other eliminations are lost.  The path is not exercised by the reference's tests (test/smooth_test.py:122-139 only takes it
for models with nv >= 60, of which it has none).  This package therefore keeps rejecting sparse models at device_put instead of
reproducing an incorrect factorisation.
"""
import sys
sys.path[:0]=['/root/repo/tests','/root/repo/oracle','/root/repo/mujoco-torch_amd']
import numpy as np, torch
import ref_harness
from mujoco_torch_amd import mjcf
import mujoco_torch_amd as mt
ref = ref_harness.load()
for xml in ("hopper","halfcheetah","humanoid"):
    lite = mjcf.from_xml_path(mt.test_data_path(xml+".xml"))
    lite.opt.jacobian = 1  # SPARSE
    mref = ref_harness.put_model(ref, lite)
    d = ref.io.make_data(mref)
    rng=np.random.RandomState(0)
    d = d.replace(qpos=d.qpos + torch.tensor(0.1*rng.randn(lite.nq)), qvel=torch.tensor(rng.randn(lite.nv)))
    d = ref.smooth.kinematics(mref, d); d = ref.smooth.com_pos(mref, d); d = ref.smooth.crb(mref, d); d = ref.smooth.factor_m(mref, d)
    Mfull = ref.support.full_m(mref, d).numpy()
    y = rng.randn(lite.nv)
    x = ref.smooth.solve_m(mref, d, torch.tensor(y)).numpy()
    x_true = np.linalg.solve(Mfull, y)
    print(xml, "nv", lite.nv, "sparse qM len", d.qM.shape, "solve_m rel err vs dense solve: %.2e" % (np.abs(x-x_true).max()/np.abs(x_true).max()))
