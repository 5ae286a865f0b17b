"""csrc/topo_dual_ur5.hpp is GENERATED (tools/gen_topology.py) from models/dual_ur5.json: the committed header must be what the
generator emits today, and the structural constants it compiles into the fused path's walk (TopoDualUr5S) must be facts of the model
-- the library re-checks them against the runtime model at irlosc_set_model and falls back to the shape-only walk otherwise
(tests/test_gpu_parity.py::test_a_model_without_the_structural_constants_runs_the_shape_only_walk)."""
import io
import json
import os
import re
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL = os.path.join("irl_control_amd", "models", "dual_ur5.json")
HEADER = os.path.join(ROOT, "irl_control_amd", "csrc", "topo_dual_ur5.hpp")


def _generate():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_topology
    buf = io.StringIO()
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        with redirect_stdout(buf):
            gen_topology.main(MODEL, "TopoDualUr5")
    finally:
        os.chdir(cwd)
    return buf.getvalue()


def test_committed_header_is_what_the_generator_emits():
    assert _generate() == open(HEADER).read()


def _array(text, struct, name):
    body = text[text.index(f"struct {struct}"):]
    m = re.search(rf"{name}\[N[BJ]\] = \{{([^}}]*)\}}", body)
    return [int(x) for x in m.group(1).split(",")]


def test_structural_constants_are_facts_of_the_model():
    text = open(HEADER).read()
    bodies = json.load(open(os.path.join(ROOT, MODEL)))["bodies"]
    hinges = [b["joint"] for b in bodies if b["joint"]]
    assert _array(text, "TopoDualUr5 ", "quat_id") == [0] * len(bodies)          # the shape-only struct claims nothing
    quat_id, pos_zero = _array(text, "TopoDualUr5S", "quat_id"), _array(text, "TopoDualUr5S", "pos_zero")
    icb_diag, ipos_zero = _array(text, "TopoDualUr5S", "icb_diag"), _array(text, "TopoDualUr5S", "ipos_zero")
    jpos_zero, axis_code = _array(text, "TopoDualUr5S", "jpos_zero"), _array(text, "TopoDualUr5S", "axis_code")
    for i, b in enumerate(bodies):
        assert quat_id[i] == int([float(x) for x in b["quat"]] == [1.0, 0.0, 0.0, 0.0])
        assert pos_zero[i] == int(all(float(x) == 0.0 for x in b["pos"]))
        assert icb_diag[i] == int([float(x) for x in b["iquat"]] == [1.0, 0.0, 0.0, 0.0])
        assert ipos_zero[i] == int(all(float(x) == 0.0 for x in b["ipos"]))
    for j, h in enumerate(hinges):
        assert jpos_zero[j] == int(all(float(x) == 0.0 for x in h["pos"]))
        ax = [float(x) for x in h["axis"]]
        if axis_code[j]:
            c = abs(axis_code[j]) - 1
            assert ax == [(1.0 if axis_code[j] > 0 else -1.0) if i == c else 0.0 for i in range(3)]
    assert sum(quat_id) == 16 and sum(pos_zero) == 7 and sum(icb_diag) == 21 and sum(jpos_zero) == 25 and all(axis_code)        # what DESIGN.md section 4.2 quotes
