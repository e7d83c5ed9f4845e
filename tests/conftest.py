import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass without a GPU: they are skipped (not passed) here
    # and FAIL on a GPU box if the HIP library is missing (see tests/gpu_util.py).
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_case(name, manifest):
    """Returns (meta, arrays) with inputs materialised (stored, shared file, or regenerated from seed)."""
    import torch
    from oracle.inputs import gen_inputs, positions, sha256_of

    meta = manifest[name]
    arrays = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if "x" not in arrays:
        if "inputs" in meta:
            arrays.update(dict(np.load(os.path.join(GOLDEN, meta["inputs"] + ".npz"))))
        else:
            B, n, m = meta["shape"]
            x, y = gen_inputs(meta["kind"], B, n, m, meta["seed"])
            assert sha256_of(x, y) == bytes(arrays["inputs_sha256"]).hex(), "seeded inputs drifted"
            spec = "linspace" if meta["pos"] == "linspace" else "rfftfreq"
            pos = positions(spec, n)
            arrays.update(x=x.numpy(), y=y.numpy(), x_pos=pos.numpy(), y_pos=pos.numpy().copy())
    if "x_pos" not in arrays:  # fixed_x form
        n = arrays["x"].shape[-1]
        arrays["x_pos"] = arrays["y_pos"] = torch.linspace(0, 1, n).numpy()
    return meta, arrays


def case_names(manifest_path=os.path.join(GOLDEN, "manifest.json")):
    with open(manifest_path) as f:
        return [k for k in json.load(f) if not k.startswith("_")]


def ctor_to_flags(ctor):
    from oracle import sot_oracle as so
    return float(ctor.get("p", 1)), so.make_flags(ctor.get("square_dist", False), ctor.get("dont_normalize", False),
                                                  ctor.get("limit_quantile_range", False),
                                                  ctor.get("require_sort", True))
