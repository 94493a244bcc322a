"""The C-ABI library loads and exports every symbol include/mjhip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from mujoco_torch_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(native.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return native.load_library()


def test_header_entry_points_exported(lib):
    text = open(os.path.join(ROOT, "include", "mjhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(mjh_\w+)\s*\(", text))
    assert {"mjh_model_create", "mjh_model_destroy", "mjh_forward", "mjh_step", "mjh_last_error"} <= names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mjhip.h but not exported"


def test_field_lists_match_header(lib):
    native.check_abi(lib)
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mjhip.h")).read()
    assert lib.mjh_abi_version() == int(re.search(r"#define MJH_ABI_VERSION (\d+)", header).group(1))


def test_struct_sizes_match_binding(lib):
    n_data = len(native.LISTS["MJH_DATA_REALS"]) + len(native.LISTS["MJH_DATA_I32"]) + len(native.LISTS["MJH_DATA_I64"]) + len(native.LISTS["MJH_DATA_EXTRA_IN"])
    assert ctypes.sizeof(native.DataPtrs) == 8 * n_data == lib.mjh_sizeof_data()


def test_oracle_exports():
    import pyoracle

    lib = pyoracle.lib()
    for n in ("mjo_step", "mjo_forward", "mjo_max_threads"):
        assert hasattr(lib, n)
