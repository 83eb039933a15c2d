import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "movie-recommender-demo_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


_ACCURACY = {}


@pytest.fixture(scope="session")
def accuracy():
    """Recorder of achieved floating-point errors: accuracy(case, engine, err_over_bound, **extra).  Dumped to
    gpurun_out/accuracy.json at session end (copied to profiles/rNN_accuracy.json by the builder), so that the
    tolerances in tests/cases.py sit next to what was actually measured."""
    def rec(case, engine, err_over_bound, **extra):
        e = _ACCURACY.setdefault(case, {}).setdefault(engine, {"max_err_over_bound": 0.0})
        e["max_err_over_bound"] = max(e["max_err_over_bound"], float(err_over_bound))
        e.update({k: (float(v) if isinstance(v, (int, float, np.floating)) else v) for k, v in extra.items()})
    yield rec


def pytest_sessionfinish(session, exitstatus):
    if not _ACCURACY:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "accuracy.json"), "w") as f:
        json.dump({"bounds": "tests/cases.py: scaled = 1e-5*max(1,|logit|); randn = max(1e-4*max(1,|logit|), 1e-5*batch scale)",
                   "cases": _ACCURACY}, f, indent=1, sort_keys=True)
