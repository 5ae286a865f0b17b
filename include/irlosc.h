/*
 * irlosc.h — C ABI of the MI355X-native batched operational-space controller (libirlosc.so).
 *
 * This is the drop-in boundary that sits UNDER irl_control's Python `OSC.generate()`:
 * the reference has no FFI of its own (it is pure Python/NumPy), so each entry point below names
 * the reference lines whose work it takes over.  Plain pointers and sizes only; no torch types.
 *
 *   reference work                                              -> entry point
 *   ----------------------------------------------------------------------------------------------
 *   OSC.__init__ gain tables (osc.py:19-39), device row masks /
 *   joint sets (device.py:36,66-69; robot.py:28-32,50-55)       -> irlosc_create, irlosc_set_gains
 *   Robot.get_all_states(): M, dq, J stack, EE pose, wrench
 *   (robot.py:44-72,125-136; device.py:115-170; osc.py:132-138) -> irlosc_upload (records assembled by the caller)
 *                                                                  irlosc_upload_raw (assembled on the GPU from raw
 *                                                                  simulator arrays)
 *   targets dict (utils.py:5-67; osc.py:156-159,172)            -> irlosc_set_targets
 *   OSC.generate numerical body (osc.py:41-118,144-200)         -> irlosc_step / irlosc_step_device
 *   forces gather u_all[actuator_trnids] (osc.py:203-210)       -> host side, from u[B,n]
 *   the per-tick MuJoCo reads themselves: mj_fullM (robot.py:68-72),
 *   jacp / jacr (device.py:115-133), qfrc_bias (osc.py:190-191),
 *   EE xpos / xquat (device.py:97-99), from (qpos, qvel)        -> irlosc_set_model, irlosc_upload_q, then irlosc_frontend
 *                                                                  (dense records) or irlosc_step_from_q (fused: no records)
 *
 * Record layouts (batch-major, row-major, element type = cfg.dtype: float or double):
 *   M[B][n][n]      joint-space inertia, symmetric positive definite (the row16 kernels read row j as column j)
 *   J[B][k][n]      stacked task Jacobian, device blocks in TARGETS order, k = sum(dev_rows)
 *   dq[B][n]        joint velocities
 *   bias[B][n]      qfrc_bias (gravity + Coriolis); ignored unless IRLOSC_USE_G
 *   ee_pose[B][ndev][7]   x y z qw qx qy qz of each target device's end effector
 *   tgt_pose[B][ndev][7]  target x y z qw qx qy qz (quaternion need not be normalised)
 *   tgt_vel[B][ndev][6]   hstack(xyz_vel, abg_vel) or NULL (= all zero -> damping branch A)
 *   wrench[B][ndev][6]    world-frame F/T sensor reading or NULL; used only with IRLOSC_ADMITTANCE
 *   u[B][n]         joint torques u_all (osc.py:152-200)
 *   flags[B]        uint32 status bits per instance (IRLOSC_FLAG_*)
 *
 * Contracts the kernels rely on (not checked on the device):
 *   - M is symmetric: the throughput kernels read row j of M as its column j (the generic kernel uses M as given, like
 *     osc.py:49,151).  Records that come from the HOST are probed on the device: irlosc_upload and irlosc_tick return
 *     IRLOSC_ERR_ARG when an instance has max |M - M^T| > 1e-6 max |M| (over its FINITE entries: a robot whose M holds
 *     NaN / Inf is not refused -- it is reported per instance through IRLOSC_FLAG_NONFINITE / M_NOT_PD and the other robots
 *     of the batch get their torques) and the context runs a throughput kernel.  Records
 *     assembled on the device (irlosc_upload_raw / irlosc_assemble_device / irlosc_frontend) are symmetric by
 *     construction; for irlosc_step_device it stays the caller's contract;
 *   - (not a contract, an observation the library makes for itself) records of a real robot carry the zeros of its kinematic
 *     tree -- robot.py:68-72 / device.py:115-133 hand over what mj_fullM / mj_jacBody wrote -- and uploads are probed for
 *     them: see irlosc_slot_structure;
 *   - device pointers handed to irlosc_step_device / irlosc_assemble_device are 16-byte aligned;
 *   - a context is driven from ONE stream at a time: its train tables, worklists and pending stage-2 work are ordered
 *     by stream order only, so a caller stream passed to the *_device entry points must not run concurrently with the
 *     context's own stream or with another caller stream on the same context;
 *   - a step over B instances needs B instances of state and of targets in the slot (IRLOSC_ERR_STATE otherwise).
 *
 * Ownership: the caller owns every host buffer (borrowed for the duration of the call); the
 * library owns its device buffers and its HIP stream.  Errors: 0 = success, negative irlosc_status
 * otherwise, message via irlosc_last_error(); nothing throws across the ABI.  A context is bound
 * to one GPU and is not thread-safe; distinct contexts may be driven from distinct threads.
 * There is NO CPU fallback: every compute entry point fails with IRLOSC_ERR_HIP without a GPU.
 */
#ifndef IRLOSC_H
#define IRLOSC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): a fused irlosc_step_from_q invalidates the slot's dense records (IRLOSC_ERR_STATE on a later irlosc_step);
 * IRLOSC_KERNEL_AUTO on float32 records means fp64 arithmetic (row16 "mixed": there is no fp32-arithmetic kernel); every n = 25 layout
 * (k <= 16, ndev <= 4) has a row16-class kernel; irlosc_kernel_class / irlosc_giveup_counts / irlosc_time_trains added;
 * only the irlosc_* symbols are exported.  _lib.py refuses a library whose version differs. */
#define IRLOSC_ABI_VERSION 3
/* the entry points below are the ONLY dynamic symbols of libirlosc.so (built with -fvisibility=hidden) */
#define IRLOSC_API __attribute__((visibility("default")))
#define IRLOSC_MAX_DEV 4
#define IRLOSC_MAX_N 32
#define IRLOSC_MAX_K 16
#define IRLOSC_GAIN_WORDS 12 /* kp kv ko k0 k1 k2 d0 d1 d2 max_vel0 max_vel1 has_max_vel */

typedef enum { IRLOSC_F32 = 0, IRLOSC_F64 = 1 } irlosc_dtype;

typedef enum {
    IRLOSC_OK = 0,
    IRLOSC_ERR_ARG = -1,     /* bad argument / layout */
    IRLOSC_ERR_HIP = -2,     /* HIP runtime error (no device, OOM, launch failure) */
    IRLOSC_ERR_STATE = -3    /* call order (e.g. step before upload) */
} irlosc_status;

/* cfg.flags */
#define IRLOSC_USE_G      1u  /* add bias forces, osc.py:190-191 */
#define IRLOSC_ADMITTANCE 2u  /* add ext_f to the task signal, osc.py:184-185 */
#define IRLOSC_NULLSPACE  4u  /* null-space damping, osc.py:195-200 */

/* per-instance status bits written by the kernels */
#define IRLOSC_FLAG_M_NOT_PD      1u   /* Cholesky of M met a non-positive pivot */
#define IRLOSC_FLAG_PINV_BRANCH   2u   /* |det(J M^-1 J^T)| < 1e-4: reference takes np.linalg.pinv (osc.py:55) */
#define IRLOSC_FLAG_EIGEN_PATH    4u   /* k x k eigen-decomposition was needed (not certifiably cond < 1e5) */
#define IRLOSC_FLAG_TRUNCATED     8u   /* at least one eigenvalue <= 1e-5 * max was dropped */
#define IRLOSC_FLAG_VEL_BRANCH_B 16u   /* some device had all six target-velocity components non-zero (osc.py:173-177) */
#define IRLOSC_FLAG_BAD_JIDX     32u   /* branch B indexed dx out of range (IndexError in the reference) */
#define IRLOSC_FLAG_NONFINITE    64u   /* output contains NaN/Inf */

/* kernel selection (cfg.kernel) */
#define IRLOSC_KERNEL_AUTO    0   /* fp64 ARITHMETIC always (the reference's, osc.py:49-55; meets 1e-5): row16 for every n = 25
                                     layout -- on float64 records and on float32 records alike -- else generic */
#define IRLOSC_KERNEL_GENERIC 1   /* one wavefront per instance, LDS tiles, any n<=32, k<=16 */
#define IRLOSC_KERNEL_REMOVED_GROUP 2 /* (ABI versions 1-2: an fp32-ARITHMETIC kernel, error ~ eps32 * cond(J M^-1 J^T) -- 14 % of physical
                                     instances missed the 1e-5 contract.  Removed in version 3: irlosc_create answers IRLOSC_ERR_ARG.) */
#define IRLOSC_KERNEL_ROW16   3   /* fp64 arithmetic: 16 lanes (one DPP row) per instance, broadcast-FMA formulation (n = 25;
                                     instantiations for (k, ndev) = (13,3), (12,2), (7,3), (6,2), KMAX-padded variants for every
                                     other k <= 16, ndev <= 4: irlosc_kernel_class); float32 records = the "mixed" path */

typedef struct irlosc_cfg {
    int32_t hip_device;                      /* HIP device ordinal */
    int32_t dtype;                           /* irlosc_dtype */
    int32_t max_batch;                       /* capacity B_max of the resident buffers */
    int32_t n_slots;                         /* >=1 resident input sets (bench rotates them to defeat the 256 MiB L3) */
    int32_t n;                               /* robot.num_joints_total (robot.py:32) */
    int32_t ndev;                            /* number of target devices, targets order */
    uint32_t flags;                          /* IRLOSC_USE_G | IRLOSC_ADMITTANCE | IRLOSC_NULLSPACE */
    int32_t kernel;                          /* IRLOSC_KERNEL_* */
    int32_t dev_rows[IRLOSC_MAX_DEV];        /* r_d = popcount(ctrlr_dof[d]) */
    uint8_t ctrlr_dof[IRLOSC_MAX_DEV][6];    /* row mask hstack(ctrlr_dof_xyz, ctrlr_dof_abg), device.py:36 */
    uint8_t calc_xyz[IRLOSC_MAX_DEV];        /* np.sum(device.ctrlr_dof_xyz) > 0, osc.py:108 */
    uint8_t calc_abg[IRLOSC_MAX_DEV];        /* np.sum(device.ctrlr_dof_abg) > 0, osc.py:113 */
    uint32_t joint_mask[IRLOSC_MAX_DEV];     /* bit j set <=> j in device.joint_ids_all (osc.py:174) */
    int32_t j_idx0[IRLOSC_MAX_DEV];          /* first row of J_idxs[name] (robot.py:50-55), used by branch B */
} irlosc_cfg;

typedef struct irlosc_ctx irlosc_ctx;

IRLOSC_API int irlosc_abi_version(void);

/* Number of HIP devices visible, or a negative irlosc_status. */
IRLOSC_API int irlosc_device_count(void);

IRLOSC_API int irlosc_create(const irlosc_cfg* cfg, irlosc_ctx** out);
IRLOSC_API void irlosc_destroy(irlosc_ctx* ctx);
IRLOSC_API const char* irlosc_last_error(const irlosc_ctx* ctx); /* ctx may be NULL: last create() error */
IRLOSC_API const char* irlosc_kernel_name(const irlosc_ctx* ctx); /* name of the kernel irlosc_step launches */
/* What irlosc_create settled on, for a C caller that asked for IRLOSC_KERNEL_AUTO and wants to know whether it got a throughput
 * kernel: IRLOSC_CLASS_GENERIC means the one-wavefront-per-instance kernel (any n <= 32: ~25x slower per instance at large batches;
 * AUTO only lands there when n != 25).  ROW16 = an instantiation for exactly this (k, ndev); ROW16_PADDED = the row16 kernel of the
 * smallest tier KMAX in {4, 7, 10, 13, 16} >= k, with k and ndev as run-time arguments (same results bit for bit, a few per cent to
 * a third slower than an exact instantiation would be; the name then ends in "_ndev<d>_pad<KMAX>").
 * One difference in HOW a result is reached, not in the result: a task row no joint can move (an exact zero row of J) is taken out of
 * the k x k factorisation up front by the padded kernels and by the lane-per-robot step of the fused path (PINV and TRUNCATED set, no
 * eigen stage needed for it), while the four exact instantiations find it in their eigen stage (three such directions at most, then the
 * give-up list -> generic kernel).  Torques agree to rounding and the flags PINV / TRUNCATED agree; irlosc_giveup_counts and the
 * throughput on such inputs differ.  No layout of the shipped examples has such a row. */
#define IRLOSC_CLASS_GENERIC      0
#define IRLOSC_CLASS_ROW16        1
#define IRLOSC_CLASS_ROW16_PADDED 2
IRLOSC_API int irlosc_kernel_class(const irlosc_ctx* ctx);
IRLOSC_API const char* irlosc_frontend_name(const irlosc_ctx* ctx); /* name of the kernel irlosc_frontend launches ("" before irlosc_set_model) */

/* gains[nb][ndev][IRLOSC_GAIN_WORDS] and null_kv[nb] as double; nb == 1 broadcasts one gain set to
 * every instance (kept in constant/SGPR space), nb == max_batch gives per-instance gains. */
IRLOSC_API int irlosc_set_gains(irlosc_ctx* ctx, const double* gains, const double* null_kv, int32_t nb);

/* Host -> device copy of one batch of robot state into resident slot `slot`.  wrench may be NULL.  On the throughput paths
 * an asymmetric M is refused (IRLOSC_ERR_ARG, the slot then holds nothing): see the contracts block above. */
IRLOSC_API int irlosc_upload(irlosc_ctx* ctx, int32_t slot, int32_t B, const void* M, const void* J,
                  const void* dq, const void* bias, const void* ee_pose, const void* wrench);
/* 1 when the records now in `slot` carry the zero pattern of the compiled Dual-UR5 tree and the fp64 row16 kernel will
 * therefore run its factorisation in the tree-structured form (M = L^T L from the leaves up, fill-in free: Featherstone's
 * branch-induced sparsity; rows of Y = L^-T J^T under no end effector skipped), 0 when it runs the dense recursion.  The
 * verdict is taken when the records arrive: irlosc_upload / irlosc_upload_raw look at every instance (one pass on the device:
 * M[i][j] exactly 0 unless hinge i is above hinge j or j above i; columns of J exactly 0 for hinges that move no end-effector
 * candidate -- what mj_fullM / mj_jacBody leave, robot.py:68-72, device.py:115-133), batches of 64 instances and more only;
 * irlosc_frontend's lane kernel writes such records by construction; irlosc_assemble_device (caller's stream) does not
 * qualify until irlosc_probe_structure has looked.  The form is chosen per slot: a train of irlosc_step_resident that mixes
 * qualifying and other slots is issued as two launches.  Results differ from the dense recursion at rounding level only.  IRLOSC_TREE=0 in the environment turns the
 * form off. */
IRLOSC_API int irlosc_slot_structure(const irlosc_ctx* ctx, int32_t slot);
/* The same look at records that are already in `slot` (first B instances), for the one path that cannot take it by itself:
 * irlosc_assemble_device on a caller's stream.  Synchronous on the context's stream -- the caller synchronises its own stream
 * first.  Returns 1 / 0 like irlosc_slot_structure (and updates that verdict), or a negative irlosc_status. */
IRLOSC_API int irlosc_probe_structure(irlosc_ctx* ctx, int32_t slot, int32_t B);
/* Host -> device copy of the targets for slot `slot`.  tgt_vel may be NULL (all zero). */
IRLOSC_API int irlosc_set_targets(irlosc_ctx* ctx, int32_t slot, int32_t B, const void* tgt_pose,
                       const void* tgt_vel);

/* One control step over the B instances of `slot`: launch, then copy u[B][n] (and flags[B], may
 * be NULL) back to the host buffers.  u_host may be NULL to leave the result on the device. */
IRLOSC_API int irlosc_step(irlosc_ctx* ctx, int32_t slot, int32_t B, void* u_host, uint32_t* flags_host);

/* Benchmark form: `iters` back-to-back steps on resident data, slot = (first_slot + i) % n_slots,
 * no host copies.  Consecutive steps are independent batches, so the throughput path chains several of them
 * into one launch and lets the eigen-path stage of a launch's steps ride in the next launch; everything is
 * complete when the call returns, and irlosc_download then yields the LAST step's outputs.  *ms_total receives
 * the HIP-event time of the whole region on the library's own stream; *ms_kernel_avg = *ms_total / iters. */
IRLOSC_API int irlosc_step_resident(irlosc_ctx* ctx, int32_t first_slot, int32_t B, int32_t iters,
                         float* ms_total, float* ms_kernel_avg);

/* Roofline support: mean duration of the DOMINANT kernel launch, measured live with one HIP event pair per launch
 * on the library's stream, over about `iters` (<= 256) steps run exactly like irlosc_step_resident.  Group path:
 * the fused launch (stage 1 of irlosc_steps_per_launch() chained steps + the riding stage 2 of the previous
 * launch's steps); generic path: the generic kernel.  Outputs are complete, as after irlosc_step_resident. */
IRLOSC_API int irlosc_time_dominant_kernel(irlosc_ctx* ctx, int32_t slot, int32_t B, int32_t iters, float* ms_avg);
/* Roofline evidence WITHOUT a tracer ("Timing method"; row16 kernel only).  `ntrains` (<= 4096) consecutive
 * trains of irlosc_steps_per_launch() steps, issued exactly as irlosc_step_resident issues them (slots rotating from first_slot),
 * after one untimed train.  Every train gets (a) its own HIP event pair on the library's stream and (b) the wall clock
 * (s_memrealtime, 100 MHz) stamped by its main kernel itself: the start of its first wave and the end of its last wave.
 *   out[4 i + 0]  event pair of train i in milliseconds (main kernel + give-up pass; from_q: walk + OSC kernel + give-up pass)
 *   out[4 i + 1]  start of train i's first wave, microseconds after the first wave of train 0
 *   out[4 i + 2]  end of train i's last wave, same origin
 *   out[4 i + 3]  shader clock in MHz during train i: cycle counter against wall clock over the life of one sample wave (the kernel is
 *                 fp64-VALU-heavy and the part power-limited: 1.5-1.8 GHz under this load against the 2.4 GHz peak, and box to box different)
 * => in-kernel duration of train i = out[4 i + 2] - out[4 i + 1] (what a kernel trace reports per dispatch, minus the
 * tracer's own serialisation); steady-state period = out[4 (i + 1) + 1] - out[4 i + 1] (what irlosc_step_resident's wall
 * time divided by the number of trains measures).  from_q != 0: trains of irlosc_step_resident_from_q (the stamps are the OSC
 * kernel's; the walk in front of it is inside the period and the event pair). */
IRLOSC_API int irlosc_time_trains(irlosc_ctx* ctx, int32_t first_slot, int32_t B, int32_t ntrains, int32_t from_q, double* out);
/* How many instances the most recent step (out[0]) / the steps of the most recent train (out[0 .. irlosc_steps_per_launch() - 1]) handed
 * from the row16 kernel's in-wave eigen stage to the generic kernel -- task spaces that lose MORE than three directions at once
 * (osc.py:55 with four or more singular values under the cut).  Results are the same either way; throughput is not: the give-up
 * pass is a serial tail of its train (one instance costs ~27 us per train, a batch dominated by them runs at 4.5e6 steps/s instead of
 * 6e8).  Zero on physical states of the Dual-UR5 in every sweep so far; a caller whose task sets are rank-deficient by construction
 * can watch this counter.  out[8]; all zero on the other kernels.  (A train that mixes tree-form and dense slots is two launches;
 * every step still reports at its own index of the train.)  Synchronises the context's stream. */
IRLOSC_API int irlosc_giveup_counts(irlosc_ctx* ctx, int32_t* out);
/* Steps chained in one launch by irlosc_step_resident / irlosc_time_dominant_kernel (1 on the generic path):
 * the algorithmic bytes of one dominant launch = this many steps' worth. */
IRLOSC_API int irlosc_steps_per_launch(const irlosc_ctx* ctx);

/* State assembly on the GPU (what Robot.get_all_states() / Device.get_state() do per robot on the host:
 * robot.py:44-72,125-136; device.py:115-170): one batch of RAW simulator arrays in, the resident records of slot
 * `slot` out (M, J, dq, bias, ee_pose, wrench), same result as irlosc_upload of host-assembled records.
 *   qM[B][nv][nv]            dense inertia matrix (mj_fullM), nv = sim.model.nv (may exceed n: free bodies)
 *   qvel[B][nv], qfrc_bias[B][nv]
 *   jacp[B][ndev][3][nv], jacr[B][ndev][3][nv]   EE-body Jacobians of the target devices, targets order
 *   ee_xpos[B][ndev][3], ee_xquat[B][ndev][4]    EE-body pose (w,x,y,z)
 *   site_xmat[B][ndev][9], sensordata[B][n_sensor]   F/T site frames and raw sensor readings; both may be NULL
 *                                                    (wrench = 0), devices without a sensor have ft_force0 = -1
 * The row mask ctrlr_dof and the block order come from the context's cfg. */
typedef struct irlosc_raw_desc {
    int32_t nv;                               /* dofs in the scene (columns of the raw Jacobians) */
    int32_t n_sensor;                         /* sensordata length per instance */
    int32_t joint_ids[IRLOSC_MAX_N];          /* robot.joint_ids_all: raw dof of robot position p (robot.py:28-32) */
    int32_t dq_src[IRLOSC_MAX_N];             /* raw dof whose qvel lands in dq[p], -1 = 0 (robot.py:60-65) */
    int32_t ft_force0[IRLOSC_MAX_DEV];        /* first sensordata index of the force triple, -1 = no sensor (device.py:139-170) */
    int32_t ft_torque0[IRLOSC_MAX_DEV];
} irlosc_raw_desc;
IRLOSC_API int irlosc_upload_raw(irlosc_ctx* ctx, int32_t slot, int32_t B, const irlosc_raw_desc* desc, const void* qM,
                      const void* qvel, const void* qfrc_bias, const void* jacp, const void* jacr,
                      const void* ee_xpos, const void* ee_xquat, const void* site_xmat, const void* sensordata);

/* The same with M as MuJoCo itself holds it: mjData.qM, the sparse lower triangle over the kinematic tree -- nM values per instance,
 * dof i's run starts at dof_Madr[i] and walks UP the tree: M[i][i], M[i][parent(i)], M[i][parent(parent(i))], ... with parent =
 * mjModel.dof_parentid (-1 ends the run).  The expansion mj_fullM does on the host (robot.py:68-72 calls it for every robot and tick)
 * runs on the GPU instead, inside the assembly kernel: 1.4 KB instead of 5 KB (nv = 25) or 11 KB (nv = 37) per robot cross PCIe.
 * qM_sparse[B][nM] in the context's dtype; everything else as irlosc_upload_raw. */
#define IRLOSC_MAX_NV 128
typedef struct irlosc_qm_layout {
    int32_t nM;                               /* mjModel.nM */
    int32_t dof_Madr[IRLOSC_MAX_NV];          /* mjModel.dof_Madr[0 .. nv) */
    int32_t dof_parentid[IRLOSC_MAX_NV];      /* mjModel.dof_parentid[0 .. nv) */
} irlosc_qm_layout;
IRLOSC_API int irlosc_upload_raw_sparse(irlosc_ctx* ctx, int32_t slot, int32_t B, const irlosc_raw_desc* desc, const irlosc_qm_layout* qml,
                             const void* qM_sparse, const void* qvel, const void* qfrc_bias, const void* jacp, const void* jacr,
                             const void* ee_xpos, const void* ee_xquat, const void* site_xmat, const void* sensordata);

/* Same assembly for simulators whose state already lives in HBM: every array pointer is a DEVICE pointer (layouts
 * as above), hip_stream a hipStream_t (NULL = the context's stream).  No copies; the call returns after enqueueing the
 * kernel, and steps of this context issued on the same stream see the assembled slot. */
IRLOSC_API int irlosc_assemble_device(irlosc_ctx* ctx, int32_t slot, int32_t B, const irlosc_raw_desc* desc, const void* d_qM,
                           const void* d_qvel, const void* d_qfrc_bias, const void* d_jacp, const void* d_jacr,
                           const void* d_ee_xpos, const void* d_ee_xquat, const void* d_site_xmat,
                           const void* d_sensordata, void* hip_stream);

IRLOSC_API int irlosc_download(irlosc_ctx* ctx, int32_t B, void* u_host, uint32_t* flags_host);
IRLOSC_API int irlosc_sync(irlosc_ctx* ctx);          /* waits for the context's stream */
IRLOSC_API int irlosc_device_sync(irlosc_ctx* ctx);   /* hipDeviceSynchronize on the context's GPU (bench bracket) */

/* One control tick in ONE call (what OSC.generate does per tick for B robots, osc.py:120-210): the records are packed
 * into a pinned staging block, cross PCIe in one copy, the step runs on them in place, u[B][n] and flags[B] come back
 * in one copy, and the call synchronises once.  Same record layouts as irlosc_upload / irlosc_set_targets; wrench and
 * tgt_vel may be NULL.  Does not touch the resident slots. */
IRLOSC_API int irlosc_tick(irlosc_ctx* ctx, int32_t B, const void* M, const void* J, const void* dq, const void* bias,
                const void* ee_pose, const void* wrench, const void* tgt_pose, const void* tgt_vel, void* u_host,
                uint32_t* flags_host);

/* Raw-device-pointer form for callers that already hold the state in HBM (e.g. an on-GPU
 * simulator): same layouts as above, all pointers are device pointers, hip_stream is a
 * hipStream_t (NULL = the context's stream).  No copies, no synchronisation. */
IRLOSC_API int irlosc_step_device(irlosc_ctx* ctx, int32_t B, const void* dM, const void* dJ, const void* ddq,
                       const void* dbias, const void* dee_pose, const void* dtgt_pose,
                       const void* dtgt_vel, const void* dwrench, void* du, uint32_t* dflags,
                       void* hip_stream);

/* ---- rigid-body front end: joint coordinates in, records assembled on the GPU -----------
 * Replaces the per-tick reads the reference makes from MuJoCo on the host - mj_fullM (robot.py:68-72), the EE-body
 * Jacobians jacp / jacr (device.py:115-133), qfrc_bias (osc.py:190-191), EE xpos / xquat (device.py:97-99) - by forward
 * kinematics, Jacobians, composite-rigid-body M and recursive-Newton-Euler bias forces computed from (qpos, qvel) for the
 * whole batch.  The model is a tree of bodies, each welded to its parent or attached by ONE hinge (MuJoCo conventions:
 * body frame pos + quat relative to the parent, hinge axis / anchor in the body frame, inertial frame ipos + iquat with
 * principal moments, quaternions w x y z).  Bodies must be listed parents first (MuJoCo's numbering does that);
 * joint j of the model is position j of the n-vector (n = cfg.n).  ee_body[d] = body whose frame is target device d's
 * end effector (targets order).  The F/T wrench is not a function of (qpos, qvel): it stays whatever irlosc_upload /
 * irlosc_upload_raw last put into the slot (absent: zero). */
#define IRLOSC_MAX_BODIES 64
typedef struct irlosc_model {
    int32_t nb;                               /* bodies, <= IRLOSC_MAX_BODIES */
    int32_t nj;                               /* hinges, must equal cfg.n */
    int32_t parent[IRLOSC_MAX_BODIES];        /* -1 = world */
    int32_t joint_of_body[IRLOSC_MAX_BODIES]; /* hinge index, or -1 = welded to the parent */
    double pos[IRLOSC_MAX_BODIES][3];
    double quat[IRLOSC_MAX_BODIES][4];
    double jaxis[IRLOSC_MAX_N][3];            /* unit axis, body frame */
    double jpos[IRLOSC_MAX_N][3];             /* anchor, body frame */
    double armature[IRLOSC_MAX_N];
    double mass[IRLOSC_MAX_BODIES];
    double ipos[IRLOSC_MAX_BODIES][3];
    double iquat[IRLOSC_MAX_BODIES][4];
    double inertia[IRLOSC_MAX_BODIES][3];     /* principal moments in the inertial frame */
    double gravity[3];
    int32_t ee_body[IRLOSC_MAX_DEV];
} irlosc_model;
/* Validates the tree (parents precede children, one body per hinge, masses >= 0) and allocates per slot qpos / qvel of
 * max_batch robots.  A model with the compiled Dual-UR5 tree shape selects the lane-per-robot kernel -- its side buffer of
 * ceil(max_batch / 64) x 266 x 512 bytes (139 MB at 65 536 robots) is allocated by the first irlosc_frontend; if that fails the
 * context drops to the generic kernel -- any other tree the generic kernel.  The fused path's walk (irlosc_step_from_q) exists twice
 * for that shape: with the structural constants of the Dual-UR5's MJCF compiled in (body frames not rotated against / coincident with
 * their parent's, hinges about coordinate axes through their body's origin, inertial frames aligned with the body frame: csrc/topo_dual_ur5.hpp,
 * TopoDualUr5S) and shape-only; the model's numbers are checked against those constants here and decide which one runs
 * (irlosc_from_q_name says: "..._dual_ur5_s + ..." is the first). */
IRLOSC_API int irlosc_set_model(irlosc_ctx* ctx, const irlosc_model* model);
/* Joint positions and velocities of one batch into resident slot `slot`: qpos[B][n], qvel[B][n], always double.  On a context whose
 * model takes the fused path the library keeps a second copy in the walk's own layout ([ceil(B / 64)][2 n][64 robots]: a hinge's pair is two
 * coalesced loads), written by a small kernel behind the copies. */
IRLOSC_API int irlosc_upload_q(irlosc_ctx* ctx, int32_t slot, int32_t B, const double* qpos, const double* qvel);
/* Run the front end on the slot's (qpos, qvel): fills its M, J, dq, bias, ee_pose records (asynchronous, context's
 * stream); irlosc_set_targets + irlosc_step then work as after irlosc_upload.  Afterwards the slot holds the records of
 * exactly B robots (an earlier, larger upload no longer vouches for the instances beyond B); the wrench of the slot stays what
 * the last irlosc_upload / irlosc_upload_raw wrote (undefined for instances beyond THAT call's B). */
IRLOSC_API int irlosc_frontend(irlosc_ctx* ctx, int32_t slot, int32_t B);
/* Copy records of slot `slot` back to the host (any pointer may be NULL): what irlosc_upload put there, or what the front
 * end / irlosc_upload_raw assembled on the GPU.  Same layouts and element type as irlosc_upload. */
IRLOSC_API int irlosc_download_records(irlosc_ctx* ctx, int32_t slot, int32_t B, void* M, void* J, void* dq, void* bias,
                            void* ee_pose);
/* One control step from joint coordinates: irlosc_upload_q + irlosc_set_targets, then this (results as after irlosc_step).
 * With the compiled Dual-UR5 tree shape and the fp64 row16 kernel this is the FUSED path: the lane-per-robot walk leaves only
 * the structural non-zeros of M and J, the bias forces and the EE poses in a compact exchange buffer (2.6 KB per robot,
 * written once, coalesced) and the OSC kernel gathers its operands from there -- the dense records of the slot are neither
 * written nor read -- except that robots the in-kernel eigen stage hands to the generic kernel get theirs from the
 * wave-per-robot front end, into the slot.  AFTER A FUSED STEP THE SLOT THEREFORE HOLDS NO RECORDS: irlosc_step,
 * irlosc_step_resident and irlosc_download_records on it fail with IRLOSC_ERR_STATE until irlosc_frontend / irlosc_upload* fills
 * it again (its joint coordinates and targets stay).  Other models / kernels: irlosc_frontend + irlosc_step (records left behind).
 * The exchange buffers (ceil(max_batch / 64) x 334 x 512 bytes each: 318 entries of the walk + IRLOSC_MAX_K rows for the gained task error
 * that a task pass between the walk and the OSC kernel leaves; 175 MB at 65 536 robots) are allocated by the first fused step: one for this call,
 * one per step of a train (8) for irlosc_step_resident_from_q; if that allocation fails the context drops to the path through
 * dense records.
 * The OSC step behind the walk (ABI version 3): for a layout whose task rows fit an instantiation -- at most six rows per arm and one on
 * the stand: every layout of the shipped examples -- and a train without target velocities it runs ONE LANE PER ROBOT on the exchange
 * buffer (tree-structured L^T L, Y, A = J M^-1 J^T, the k x k factorisation, certificate, solve and torques, with part 1 of the task signal
 * computed in the same kernel); the robots whose solve is a truncated pseudo-inverse (osc.py:55; ~15 % of physical states) leave a 1.5 KB
 * record and are finished by an eigen pass -- one lane per robot too where a step flags 3 000 robots or more, four records per wave on
 * thinner lists, chosen on the device per step --, what that gives up on goes to the generic kernel as before.  Everything else takes
 * the row16 FROMQ kernel (behind a task pass).  Same contract either way (1e-5 on the parity domain, same flags); irlosc_from_q_name
 * says which.  The lane form's records -- max_batch (rounded up to 64) x 1 536 bytes per step of a train -- are allocated with the
 * exchange buffers; if that fails only the lane form is switched off.  Environment switches for A/B measurements: INTEGRATION.md. */
IRLOSC_API int irlosc_step_from_q(irlosc_ctx* ctx, int32_t slot, int32_t B, void* u_host, uint32_t* flags_host);
/* What irlosc_step_from_q / irlosc_step_resident_from_q launch ("" before irlosc_set_model). */
IRLOSC_API const char* irlosc_from_q_name(const irlosc_ctx* ctx);
/* Benchmark form of the whole path from joint coordinates: `iters` steps like irlosc_step_from_q on resident (qpos, qvel),
 * chained into trains on the fused path, slot = (first_slot + i) % n_slots; HIP-event time of the region on the library's
 * stream. */
IRLOSC_API int irlosc_step_resident_from_q(irlosc_ctx* ctx, int32_t first_slot, int32_t B, int32_t iters, float* ms_total,
                                float* ms_step_avg);

/* ---- multi-GPU: the final throughput reduction -----------------------------------------------
 * Instances are independent (osc.py:120-210 touches one robot), so a node runs one process per GPU on its own shard
 * and NOTHING is exchanged per tick.  RCCL (over xGMI) is used once per benchmark: sum of the steps done, max of the
 * elapsed time, and an all-gather of per-rank output checksums.  librccl is loaded on first use (dlopen), so a
 * single-GPU deployment does not depend on it.  The 128-byte unique id is made by rank 0 (irlosc_comm_unique_id) and
 * handed to the other ranks by the host program (irl_control_amd/sharding.py: a file next to MASTER_PORT). */
typedef struct irlosc_comm irlosc_comm;
#define IRLOSC_COMM_ID_BYTES 128
IRLOSC_API int irlosc_comm_unique_id(uint8_t id_out[IRLOSC_COMM_ID_BYTES]);
IRLOSC_API int irlosc_comm_create(int32_t hip_device, int32_t rank, int32_t world, const uint8_t id[IRLOSC_COMM_ID_BYTES],
                       irlosc_comm** out);
IRLOSC_API void irlosc_comm_destroy(irlosc_comm* comm);
IRLOSC_API const char* irlosc_comm_last_error(const irlosc_comm* comm);   /* comm may be NULL: last create()/unique_id() error */
/* In place: *steps_sum <- sum over ranks, *elapsed_max <- max over ranks (two ncclAllReduce on the comm's stream,
 * then a stream sync).  Doubles as the barrier of the benchmark bracket. */
IRLOSC_API int irlosc_bench_allreduce(irlosc_comm* comm, double* steps_sum, double* elapsed_max);
/* all[r] <- rank r's `mine` (ncclAllGather): per-shard output checksums, to show that sharding changes no bit. */
IRLOSC_API int irlosc_comm_allgather_u64(irlosc_comm* comm, uint64_t mine, uint64_t* all);

#ifdef __cplusplus
}
#endif
#endif /* IRLOSC_H */
