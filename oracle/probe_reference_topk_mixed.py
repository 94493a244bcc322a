"""Container-only probe (needs /root/reference): what the REFERENCE does with ``max_contact_points`` when the candidate contacts have
mixed condims.  ``make_condim`` (collision_driver.py:618-644) sizes the constraint rows from the ``max_contact_points`` SMALLEST condims
of the candidate list, while ``collision`` (:822-840) keeps the closest contacts whatever their condims are and derives ``efc_address``
from THOSE.  The probe steps the capsules_topk model with one geom switched to condim 1 and prints, per environment, the condims of the
kept contacts, the rows they need, the static ``nefc`` the reference allocated, and what the step did.

    python oracle/probe_reference_topk_mixed.py        (profiles/r03/reference_topk_mixed_probe.txt)
"""
import os
import sys
import traceback

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "mujoco-torch_amd"))
import ref_harness  # noqa: E402
from mujoco_torch_amd import mjcf  # noqa: E402

ref = ref_harness.load()
lite = mjcf.from_xml_path(os.path.join(REPO, "mujoco-torch_amd", "mujoco_torch_amd", "test_data", "capsules_topk.xml"))
lite.geom_condim[4] = 1  # the sphere: frictionless (its pairs take max(condim) of the two geoms: with the condim-3 plane and capsules they stay 3 ...)
lite.geom_condim[0] = 1  # ... so the plane too: plane-sphere becomes condim 1, everything touching a capsule stays condim 3
mref = ref_harness.put_model(ref, lite)
ne, nf, nl, ncon, nefc = mref.constraint_sizes_py
print(f"static sizes: ncon {ncon} nefc {nefc}; make_condim -> {ref.collision_driver.make_condim(mref).tolist()}")
rng = np.random.RandomState(0)
for env in range(4):
    d = ref.io.make_data(mref)
    q = d.qpos.clone()
    q += torch.tensor(0.02 * rng.randn(q.numel()))
    if env >= 2:
        q[7 * 3 + 2] += 1.0  # environments 2, 3: the sphere lifted a metre -- its plane contact (the only condim-1 candidate) is no longer among the closest five
    d = d.replace(qpos=q)
    try:
        d = ref.forward.step(mref, d)
        dims = d.contact.contact_dim.tolist()
        rows = sum(1 if k == 1 else 2 * (k - 1) for k in dims)
        print(f"env {env}: kept condims {dims} need {rows} contact rows; efc_address {d.contact.efc_address.tolist()}; efc_J rows {d.efc_J.shape[0]} (static nefc {nefc}); finite {bool(torch.isfinite(d.qacc).all())}")
    except Exception as ex:  # noqa: BLE001
        print(f"env {env}: step raised {type(ex).__name__}: {str(ex)[:300]}")
        traceback.print_exc(limit=3)
