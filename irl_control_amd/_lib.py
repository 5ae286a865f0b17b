"""ctypes binding of libirlosc.so (include/irlosc.h).  Loads the in-tree library; there is no
fallback: if the HIP library is missing or no GPU is present, compute calls raise."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IRLOSC_LIB", os.path.join(_HERE, "libirlosc.so"))   # override: A/B builds only

MAX_DEV, MAX_N, MAX_K, GAIN_WORDS, MAX_BODIES = 4, 32, 16, 12, 64
F32, F64 = 0, 1
USE_G, ADMITTANCE, NULLSPACE = 1, 2, 4
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_ROW16 = 0, 1, 3      # (2: the fp32-arithmetic kernel removed in ABI version 3)
FLAG_M_NOT_PD, FLAG_PINV_BRANCH, FLAG_EIGEN_PATH, FLAG_TRUNCATED = 1, 2, 4, 8
FLAG_VEL_BRANCH_B, FLAG_BAD_JIDX, FLAG_NONFINITE = 16, 32, 64

EXPORTS = ["irlosc_abi_version", "irlosc_device_count", "irlosc_create", "irlosc_destroy",
           "irlosc_last_error", "irlosc_kernel_name", "irlosc_frontend_name", "irlosc_set_gains", "irlosc_upload",
           "irlosc_set_targets", "irlosc_step", "irlosc_step_resident", "irlosc_download",
           "irlosc_sync", "irlosc_step_device", "irlosc_time_dominant_kernel",
           "irlosc_steps_per_launch", "irlosc_upload_raw", "irlosc_upload_raw_sparse", "irlosc_assemble_device", "irlosc_device_sync",
           "irlosc_tick", "irlosc_comm_unique_id", "irlosc_comm_create", "irlosc_comm_destroy",
           "irlosc_comm_last_error", "irlosc_bench_allreduce", "irlosc_comm_allgather_u64", "irlosc_set_model",
           "irlosc_upload_q", "irlosc_frontend", "irlosc_step_resident_from_q", "irlosc_download_records",
           "irlosc_step_from_q", "irlosc_from_q_name", "irlosc_slot_structure", "irlosc_probe_structure", "irlosc_time_trains", "irlosc_giveup_counts",
           "irlosc_kernel_class"]
ABI_VERSION = 3
CLASS_GENERIC, CLASS_ROW16, CLASS_ROW16_PADDED = 0, 1, 2
COMM_ID_BYTES = 128


class RawDesc(C.Structure):
    """struct irlosc_raw_desc (include/irlosc.h)."""
    _fields_ = [("nv", C.c_int32), ("n_sensor", C.c_int32), ("joint_ids", C.c_int32 * 32),
                ("dq_src", C.c_int32 * 32), ("ft_force0", C.c_int32 * 4), ("ft_torque0", C.c_int32 * 4)]


MAX_NV = 128


class QmLayout(C.Structure):
    """struct irlosc_qm_layout (include/irlosc.h): MuJoCo's sparse form of M -- mjModel.nM, dof_Madr, dof_parentid."""
    _fields_ = [("nM", C.c_int32), ("dof_Madr", C.c_int32 * MAX_NV), ("dof_parentid", C.c_int32 * MAX_NV)]


class Model(C.Structure):
    """struct irlosc_model (include/irlosc.h)."""
    _fields_ = [("nb", C.c_int32), ("nj", C.c_int32), ("parent", C.c_int32 * MAX_BODIES),
                ("joint_of_body", C.c_int32 * MAX_BODIES), ("pos", (C.c_double * 3) * MAX_BODIES),
                ("quat", (C.c_double * 4) * MAX_BODIES), ("jaxis", (C.c_double * 3) * MAX_N),
                ("jpos", (C.c_double * 3) * MAX_N), ("armature", C.c_double * MAX_N), ("mass", C.c_double * MAX_BODIES),
                ("ipos", (C.c_double * 3) * MAX_BODIES), ("iquat", (C.c_double * 4) * MAX_BODIES),
                ("inertia", (C.c_double * 3) * MAX_BODIES), ("gravity", C.c_double * 3), ("ee_body", C.c_int32 * MAX_DEV)]


class Cfg(C.Structure):
    _fields_ = [("hip_device", C.c_int32), ("dtype", C.c_int32), ("max_batch", C.c_int32),
                ("n_slots", C.c_int32), ("n", C.c_int32), ("ndev", C.c_int32),
                ("flags", C.c_uint32), ("kernel", C.c_int32),
                ("dev_rows", C.c_int32 * MAX_DEV),
                ("ctrlr_dof", (C.c_uint8 * 6) * MAX_DEV),
                ("calc_xyz", C.c_uint8 * MAX_DEV), ("calc_abg", C.c_uint8 * MAX_DEV),
                ("joint_mask", C.c_uint32 * MAX_DEV), ("j_idx0", C.c_int32 * MAX_DEV)]


class IrloscError(RuntimeError):
    pass


_lib = None


def load():
    """Load libirlosc.so (built by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IrloscError(f"{LIB_PATH} not built: run `python __graft_entry__.py` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int32
    lib.irlosc_abi_version.restype = C.c_int
    if lib.irlosc_abi_version() != ABI_VERSION:
        raise IrloscError(f"{LIB_PATH} has ABI version {lib.irlosc_abi_version()}, this binding is written for {ABI_VERSION}: "
                          "rebuild with `python __graft_entry__.py`")
    lib.irlosc_device_count.restype = C.c_int
    lib.irlosc_create.argtypes = [C.POINTER(Cfg), C.POINTER(vp)]
    lib.irlosc_destroy.argtypes = [vp]
    lib.irlosc_destroy.restype = None
    lib.irlosc_last_error.argtypes = [vp]
    lib.irlosc_last_error.restype = C.c_char_p
    lib.irlosc_kernel_name.argtypes = [vp]
    lib.irlosc_kernel_name.restype = C.c_char_p
    lib.irlosc_kernel_class.argtypes = [vp]
    lib.irlosc_kernel_class.restype = C.c_int
    lib.irlosc_frontend_name.argtypes = [vp]
    lib.irlosc_frontend_name.restype = C.c_char_p
    lib.irlosc_set_gains.argtypes = [vp, vp, vp, i32]
    lib.irlosc_upload.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.irlosc_set_targets.argtypes = [vp, i32, i32, vp, vp]
    lib.irlosc_step.argtypes = [vp, i32, i32, vp, vp]
    lib.irlosc_step_resident.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.irlosc_time_dominant_kernel.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_float)]
    lib.irlosc_steps_per_launch.argtypes = [vp]
    lib.irlosc_time_trains.argtypes = [vp, i32, i32, i32, i32, vp]
    lib.irlosc_giveup_counts.argtypes = [vp, vp]
    lib.irlosc_upload_raw.argtypes = [vp, i32, i32, C.POINTER(RawDesc)] + [vp] * 9
    lib.irlosc_upload_raw_sparse.argtypes = [vp, i32, i32, C.POINTER(RawDesc), C.POINTER(QmLayout)] + [vp] * 9
    lib.irlosc_assemble_device.argtypes = [vp, i32, i32, C.POINTER(RawDesc)] + [vp] * 10
    lib.irlosc_download.argtypes = [vp, i32, vp, vp]
    lib.irlosc_sync.argtypes = [vp]
    lib.irlosc_step_device.argtypes = [vp, i32] + [vp] * 11
    lib.irlosc_device_sync.argtypes = [vp]
    lib.irlosc_set_model.argtypes = [vp, C.POINTER(Model)]
    lib.irlosc_upload_q.argtypes = [vp, i32, i32, vp, vp]
    lib.irlosc_frontend.argtypes = [vp, i32, i32]
    lib.irlosc_download_records.argtypes = [vp, i32, i32] + [vp] * 5
    lib.irlosc_step_resident_from_q.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.irlosc_step_from_q.argtypes = [vp, i32, i32, vp, vp]
    lib.irlosc_from_q_name.argtypes = [vp]
    lib.irlosc_from_q_name.restype = C.c_char_p
    lib.irlosc_slot_structure.argtypes = [vp, C.c_int32]
    lib.irlosc_slot_structure.restype = C.c_int
    lib.irlosc_probe_structure.argtypes = [vp, C.c_int32, C.c_int32]
    lib.irlosc_probe_structure.restype = C.c_int
    lib.irlosc_tick.argtypes = [vp, i32] + [vp] * 10
    lib.irlosc_comm_unique_id.argtypes = [vp]
    lib.irlosc_comm_create.argtypes = [i32, i32, i32, vp, C.POINTER(vp)]
    lib.irlosc_comm_destroy.argtypes = [vp]
    lib.irlosc_comm_destroy.restype = None
    lib.irlosc_comm_last_error.argtypes = [vp]
    lib.irlosc_comm_last_error.restype = C.c_char_p
    lib.irlosc_bench_allreduce.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.irlosc_comm_allgather_u64.argtypes = [vp, C.c_uint64, vp]
    for name in EXPORTS:
        getattr(lib, name)
    _lib = lib
    return lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def np_dtype(dtype_code):
    return np.float64 if dtype_code == F64 else np.float32
