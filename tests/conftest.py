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


@pytest.fixture(scope="session")
def golden():
    """Loader of the committed golden vectors (tests/golden/, produced by make_golden.py from the reference)."""
    cache = {}

    def load(name):
        if name not in cache:
            path = os.path.join(GOLDEN, name)
            if name.endswith(".json"):
                with open(path) as fh:
                    cache[name] = json.load(fh)
            else:
                cache[name] = dict(np.load(path))
        return cache[name]
    return load


def npz_json(arr):
    return json.loads(bytes(arr.tolist()).decode())


def geo(mn, mx, L):
    b = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
    return [int(1 + np.floor(mn * (b ** l))) for l in range(L)]


def table_layout(res, bw, dim):
    sizes = [min(2 ** bw, r ** dim) for r in res]
    first = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    return sizes, first, int(sum(sizes))


# named configurations of SURVEY.md section 8: (dim, resolutions, bitwidth)
CONFIGS = {
    "A": (2, geo(16, 512, 8), 11),
    "B": (2, geo(16, 512, 16), 11),
    "Bp": (2, geo(16, 2048, 16), 19),
    "D": (3, geo(16, 2048, 16), 19),
}
