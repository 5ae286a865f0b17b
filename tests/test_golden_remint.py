"""The committed fixtures ARE what the reference produces: where /root/reference exists (the build container; never the GPU box),
oracle/make_golden.py is run again into a scratch directory and every array must equal the committed one.  And the stand-in for the
reference's absent `transforms3d` dependency (oracle/shims/transforms3d) must not reach into the product: the only thing goldens and
product share is the reference."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_transforms3d_shim_is_independent_of_the_product():
    for path in glob.glob(os.path.join(ROOT, "oracle", "shims", "transforms3d", "**", "*.py"), recursive=True):
        with open(path) as f:
            code = [ln for ln in f if ln.lstrip().startswith(("import ", "from "))]
        assert not any("irl_control_amd" in ln for ln in code), (path, code)


@pytest.mark.skipif(not os.path.isdir("/root/reference/irl_control"), reason="the reference only exists in the build container")
def test_reminted_fixtures_equal_the_committed_ones(tmp_path):
    env = dict(os.environ, IRLOSC_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py")], check=True, env=env, capture_output=True, timeout=600)
    committed = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    assert len(committed) >= 29 and sorted(os.path.basename(p) for p in committed) == sorted(os.listdir(tmp_path))
    for path in committed:
        a, b = np.load(path), np.load(os.path.join(tmp_path, os.path.basename(path)))
        assert sorted(a.files) == sorted(b.files), path
        for k in a.files:
            if a[k].dtype.kind == "f":
                assert a[k].shape == b[k].shape, (path, k)
                scale = max(1.0, float(np.nanmax(np.abs(a[k])))) if a[k].size else 1.0
                assert np.nanmax(np.abs(a[k] - b[k]), initial=0.0) <= 1e-15 * scale, (path, k)      # (same code, same box: identical)
            else:
                assert np.array_equal(a[k], b[k]), (path, k)
