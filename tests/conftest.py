import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load a golden fixture -> (arrays dict, params dict of torch tensors, grads dict)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    arrs, params, grads, after = {}, {}, {}, {}
    for k in z.files:
        if k.startswith("param."):
            params[k[6:]] = torch.from_numpy(z[k])
        elif k.startswith("grad."):
            grads[k[5:]] = torch.from_numpy(z[k])
        elif k.startswith("after."):
            after[k[6:]] = torch.from_numpy(z[k])
        else:
            arrs[k] = z[k]
    if after:
        arrs["after"] = after
    return arrs, params, grads


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()
