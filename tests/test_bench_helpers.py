"""bench.py's host-side helpers that never touch the GPU: CPU accounting of the baseline leg and the NUMA placement of a rank (CPU-only tests)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # (__name__ != "__main__": no argument parsing, no launch)
    return mod


def test_cpulist_parsing():
    b = _bench()
    assert b._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert b._cpulist("") == [] and b._cpulist("5") == [5]


def test_usable_cpus_is_bounded_by_the_affinity_mask_and_the_quota():
    """the all-threads CPU leg must be sized by what this process may run on (affinity mask, cgroup quota), not by the machine's hardware threads."""
    b = _bench()
    u = b.usable_cpus()
    aff = len(os.sched_getaffinity(0))
    assert u["affinity_cpus"] == aff and 1 <= u["threads"] <= aff
    if u["cgroup_cpu_quota"] is not None:
        assert u["threads"] <= max(1, int(u["cgroup_cpu_quota"] + 1e-9))
    if u["physical_cores_in_mask"] is not None:
        assert 1 <= u["physical_cores_in_mask"] <= aff
    keep = os.sched_getaffinity(0)
    try:  # a narrower mask narrows the leg
        one = {sorted(keep)[0]}
        os.sched_setaffinity(0, one)
        assert b.usable_cpus()["threads"] == 1
    finally:
        os.sched_setaffinity(0, keep)


def test_rank_pinning_without_a_gpu_topology_changes_nothing():
    """No KFD topology in this container: the placement hint reports why it did nothing and leaves the mask alone (it must never cost a run)."""
    b = _bench()
    keep = os.sched_getaffinity(0)
    info = b.pin_rank_to_its_gpus_cores(0, 1)
    try:
        assert "pinned" in info and info["cpus_before"] == len(keep)
        if not os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
            assert info["pinned"] is False and "note" in info and os.sched_getaffinity(0) == keep
        else:
            assert (not info["pinned"]) or 1 <= info["cpus"] <= len(keep)
    finally:
        os.sched_setaffinity(0, keep)


def test_survey_bytes_table_matches_the_workloads():
    b = _bench()
    for key, wl in b.WORKLOADS.items():
        d = "f64" if str(wl["dtype"]).endswith("float64") else "f32"
        assert (key, d) in b.SURVEY_8D_BYTES, key
