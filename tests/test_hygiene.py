"""Structural checks on the test suite and the product tree (CPU)."""
import ast
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _functions_passing(path, keyword):
    tree = ast.parse(open(path).read())
    hits = set()
    for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))]:
        for call in [n for n in ast.walk(fn) if isinstance(n, ast.Call)]:
            if any(k.arg == keyword for k in call.keywords):
                hits.add(fn.name)
    return hits


def test_tail_rules_stay_with_the_pinned_outliers():
    """`check_against_oracle(..., tail_rules=True)` (the campaign's tail: ulp / ulp_band / f64 / cost / frame_cond acceptances) is triage for the
    builder-run campaign and the environments pinned from it.  No golden, seeded or in-suite campaign test may pass it."""
    allowed = {"test_gpu_parity.py": {"test_pinned_campaign_outliers"}, "test_oracle_golden.py": {"test_campaign_tail_rules_on_recorded_outputs",
                                                                                                   # (passes it only inside pytest.raises: the pinned round-5 environment must be REJECTED with the rules on too)
                                                                                                   "test_float32_accuracy_outliers_on_recorded_outputs"}}
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".py") and f not in ("_util.py", "test_hygiene.py"):
            assert _functions_passing(os.path.join(HERE, f), "tail_rules") <= allowed.get(f, set()), f


def test_the_product_tree_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import it (or the test stand-in, the reference stubs, subprocess), and the only
    dynamic library the package loads is its own libmjhip.so (native.py)."""
    pkg = os.path.join(ROOT, "mujoco-torch_amd", "mujoco_torch_amd")
    banned = re.compile(r"^(pyoracle|mjoracle|_hostsim|ref_stubs|subprocess|mujoco_torch)(\.|$)")
    loaders = {}
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            for n in ast.walk(tree):
                names = [a.name for a in n.names] if isinstance(n, ast.Import) else ([n.module or ""] if isinstance(n, ast.ImportFrom) and n.level == 0 else [])
                for name in names:
                    assert not banned.match(name), f"{f}:{n.lineno}: imports {name}"
                if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and n.func.attr in ("CDLL", "LoadLibrary", "PyDLL", "dlopen"):
                    loaders.setdefault(f, []).append(n.lineno)
    assert set(loaders) == {"native.py"} and len(loaders["native.py"]) == 1, loaders
