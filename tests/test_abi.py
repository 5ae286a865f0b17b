"""The C-ABI shared library loads and exports every symbol include/irlosc.h declares (CPU box:
no compute calls).  Compute without a GPU must fail loudly, never fall back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from irl_control_amd import _lib
from irl_control_amd.layout import OSCLayout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "irlosc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(irlosc_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 14
    for nm in names:
        assert hasattr(lib, nm), nm
    assert sorted(names) == sorted(_lib.EXPORTS)
    assert lib.irlosc_abi_version() == _lib.ABI_VERSION == 3
    hdr = open(os.path.join(ROOT, "include", "irlosc.h")).read()
    assert f"#define IRLOSC_ABI_VERSION {_lib.ABI_VERSION}" in hdr


def test_library_exports_nothing_but_the_c_abi():
    """`nm -D`: the dynamic symbols the library DEFINES are exactly the irlosc_* entry points of include/irlosc.h -- no mangled
    launch helpers, kernel handles or template instantiations (-fvisibility=hidden + the linker version script csrc/irlosc.map)."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert syms == sorted(_declared()), sorted(set(syms) ^ set(_declared()))


def test_cfg_struct_matches_header_layout():
    # 8 int32 + 4 int32 + 24 u8 + 4 u8 + 4 u8 + 4 u32 + 4 i32
    assert C.sizeof(_lib.Cfg) == 8 * 4 + 16 + 24 + 4 + 4 + 16 + 16


def test_raw_desc_struct_matches_header_layout():
    # struct irlosc_raw_desc: nv, n_sensor, joint_ids[32], dq_src[32], ft_force0[4], ft_torque0[4] (all int32)
    assert C.sizeof(_lib.RawDesc) == 4 * (2 + 32 + 32 + 4 + 4)
    hdr = open(os.path.join(ROOT, "include", "irlosc.h")).read()
    assert "#define IRLOSC_MAX_N 32" in hdr and "#define IRLOSC_MAX_DEV 4" in hdr


def _layout():
    return OSCLayout(n=25, dev_names=["a", "b"], ctrlr_dof=[[True] * 6, [True] * 6],
                     joint_ids=[list(range(1, 13)), list(range(13, 25))], j_idx0=[1, 7])


def test_create_validates_arguments():
    lib = _lib.load()
    cfg = _layout().to_cfg(_lib.F64, 4)
    cfg.dev_rows[0] = 5                      # inconsistent with the row mask
    h = C.c_void_p()
    assert lib.irlosc_create(C.byref(cfg), C.byref(h)) == -1
    assert b"dev_rows" in lib.irlosc_last_error(None)
    cfg = _layout().to_cfg(_lib.F64, 4)
    cfg.dtype = 7
    assert lib.irlosc_create(C.byref(cfg), C.byref(h)) == -1


def test_no_cpu_fallback_without_gpu():
    lib = _lib.load()
    if lib.irlosc_device_count() > 0:
        pytest.skip("a GPU is present")
    from irl_control_amd import BatchedOSC
    with pytest.raises(_lib.IrloscError, match="no HIP device"):
        BatchedOSC(_layout(), 4)


def test_layout_rejects_out_of_range_joint_ids():
    lay = OSCLayout(n=12, dev_names=["a"], ctrlr_dof=[[True] * 6], joint_ids=[[3, 12]], j_idx0=[0])
    with pytest.raises(ValueError):
        lay.to_cfg(_lib.F64, 1)


def test_pack_gains_shapes():
    from irl_control_amd.layout import pack_gains
    lay = _layout()
    g, nk, nb = pack_gains(lay, [200, 200], [50, 50], [200, 200], [[1, 2, 3]] * 2, [[.5, 1, 1]] * 2,
                           [[1, 5]] * 2, 10.0)
    assert g.shape == (1, 2, 12) and nb == 1 and nk.shape == (1,)
    assert list(g[0, 0]) == [200, 50, 200, 1, 2, 3, .5, 1, 1, 1, 5, 1]
    g, nk, nb = pack_gains(lay, np.full((4, 2), 200.), np.full((4, 2), 50.), np.full((4, 2), 200.),
                           [[1, 2, 3]] * 2, [[.5, 1, 1]] * 2, [[1, 5]] * 2, np.full(4, 10.0))
    assert g.shape == (4, 2, 12) and nb == 4


def test_comm_entry_points_validate_and_need_a_gpu():
    """irlosc_comm_* / irlosc_bench_allreduce (the RCCL throughput reduction): argument checks work without a GPU and
    creating a communicator without one fails loudly."""
    lib = _lib.load()
    h = C.c_void_p()
    idbuf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    assert lib.irlosc_comm_create(0, 2, 2, idbuf, C.byref(h)) == -1          # rank outside world
    assert b"rank" in lib.irlosc_comm_last_error(None)
    assert lib.irlosc_comm_create(0, 0, 1, None, C.byref(h)) == -1
    if lib.irlosc_device_count() < 1:
        assert lib.irlosc_comm_create(0, 0, 1, idbuf, C.byref(h)) == -2
        assert b"no HIP device" in lib.irlosc_comm_last_error(None)
    s, e = C.c_double(1.0), C.c_double(1.0)
    assert lib.irlosc_bench_allreduce(None, C.byref(s), C.byref(e)) == -1
    lib.irlosc_comm_destroy(None)


def test_model_struct_matches_header_layout():
    # struct irlosc_model: 2 int32, 2 x int32[64], pos[64][3], quat[64][4], jaxis[32][3], jpos[32][3], armature[32],
    # mass[64], ipos[64][3], iquat[64][4], inertia[64][3], gravity[3], ee_body[4]
    expect = 4 * (2 + 64 + 64) + 8 * (64 * 3 + 64 * 4 + 32 * 3 + 32 * 3 + 32 + 64 + 64 * 3 + 64 * 4 + 64 * 3 + 3) + 4 * 4
    assert C.sizeof(_lib.Model) == expect
    from irl_control_amd.rigid_body import RigidBodyModel
    m = RigidBodyModel.load("dual_ur5")
    st = m.to_struct(["ur_EE_ur5right", "ur_EE_ur5left", "ur_stand_dummy"])
    assert st.nb == 35 and st.nj == 25 and st.parent[0] == -1 and st.joint_of_body[1] == 0
    assert list(st.ee_body)[:3] == [m.body_id("ur_EE_ur5right"), m.body_id("ur_EE_ur5left"), m.body_id("ur_stand_dummy")]
