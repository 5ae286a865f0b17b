import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files if k != "layout_json"}
    g["layout"] = json.loads(str(z["layout_json"]))
    return g


def golden_gains(g):
    """gains dict (per-instance arrays) in the form oracle.generate_batch / BatchedOSC accept."""
    return dict(kp=g["kp"], kv=g["kv"], ko=g["ko"], k=g["kk"], d=g["dd"], max_vel=g["max_vel"],
                null_kv=g["null_kv"])


def golden_expected_u(g):
    """Scatter the reference's per-device forces back into the n-vector; NaN where no actuator."""
    lay = g["layout"]
    B = g["M"].shape[0]
    u = np.full((B, lay["n"]), np.nan)
    off = 0
    for trn in lay["actuator_trnids"]:
        u[:, trn] = g["forces_flat"][:, off:off + len(trn)]
        off += len(trn)
    return u


@pytest.fixture(scope="session")
def goldens():
    return {n: load_golden(n) for n in golden_names()}
