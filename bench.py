#!/usr/bin/env python3
"""bench.py — OSC control steps/sec on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f64|mixed] [--batch B] [--layout k13|k7|k12_admit]

A "step" is one pass of the OSC hot path over one resident batch of B synthetic Dual-UR5 instances per GPU
(default: B = 65 536, n = 25 joints, layout k13 = both arms xyz+abg and the base yaw, gravity and null-space
terms on: BASELINE.json configs[2]).  Inputs are resident in HBM before the timed region; `--slots` distinct
batches (default 16) are rotated so that neither successive launches nor the two trains of 8 steps that are in flight at a time
(consecutive trains run on two streams) re-hit a batch in the 256 MiB Infinity Cache.

--workload names where the dense records (M, J, dq, bias, EE poses: 8 536 B per instance in float64) come from:
  physical   (default) SURVEY.md section 8d's "true CRBA on random qpos": uniformly random joint states of the Dual-UR5, their
             records computed ONCE by the rigid-body front end before the timed region, targets scattered around the end
             effectors.  Such records carry the zeros of the kinematic tree (the two arms do not couple in M, the gripper
             joints move no end effector) exactly as MuJoCo's mj_fullM / mj_jacBody leave them, the library verifies
             that when records arrive, and the fp64 row16 kernel then factors M in the tree-structured form.
  synthetic  the recipe of the same section for when there is no front end: a random dense SPD M (no structural zeros),
             random J with the physical column pattern.  The headline of rounds 1-3; still reported under "secondary".

--dtype names the ARITHMETIC of the measured path:
  f64    float64 records, float64 arithmetic (osc_row16 kernel)  - the reference's precision, meets north_star's 1e-5
  mixed  float32 records, float64 arithmetic (osc_row16 kernel)  - BASELINE configs[2]'s fp32 storage at the 1e-5 bar
The JSON line's "dtype" is the arithmetic type ("f64" for f64 and mixed); config.records names the storage.

Instances shard across GPUs with no data-path collective (weak scaling: B per GPU is fixed; `--total-batch T`
divides T over the ranks instead, e.g. BASELINE configs[3]: --gpus 8 --total-batch 262144).  One process per GPU:
either launched as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (the launcher only
provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), or -- plain `python bench.py --gpus N` with no launcher
environment -- bench.py starts its N workers itself.  No PyTorch here: the barrier, the sum of steps, the max of the elapsed
time and the all-gather of per-rank output checksums are RCCL calls behind the C ABI (irlosc_bench_allreduce,
irlosc_comm_allgather_u64), and the device synchronisation is hipDeviceSynchronize (irlosc_device_sync).
Rank 0 prints ONE JSON line.

--require-rccl: exit non-zero unless the barrier / reduction of an N > 1 run went through RCCL on every rank (by default a rank
that cannot bring RCCL up makes ALL ranks fall back to files together, and the line says so in config.sharding /
config.rccl_ranks = 0).  --slices S: the rank's batch is the concatenation of S independently seeded sub-batches ("virtual ranks"
rank * S + j), and the line carries one output checksum per sub-batch (slice_checksums): an 8-rank run and a 1-rank
`--slices 8` run of the same total batch must agree slice by slice -- sharding changes no bit.
"""
import argparse
import json
import multiprocessing as mp
import os
import subprocess
import sys
import time

# The CPU baseline runs one oracle worker per core: the BLAS / OpenMP pools must be single-threaded BEFORE NumPy loads
# its BLAS (threadpoolctl after the fact leaves the library's 256 spinning threads per worker behind).
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 lanes x 2 flop (FMA) x 2.4 GHz: v_fma_f64 issues at the full VALU rate
# Useful fp64 flops of one control step from joint coordinates (FMA = 2; profiles/NOTES.md section 4.6 derives both figures):
FLOPS_OSC_STEP = {"k13": 23.3e3, "k12_admit": 21.6e3, "k7": 14.2e3}    # Cholesky, substitution, J M^-1 J^T, k x k, torques


def flops_osc_step(layout_name, k):
    """Counted figures for the three layouts of NOTES.md section 4.6; any other layout: the line through the k = 7 and k = 13 counts
    (1.52 kflop per task row: substitution + A = Y^T Y + k x k grow with k, the Cholesky of M does not) -- labelled as such."""
    if layout_name in FLOPS_OSC_STEP:
        return FLOPS_OSC_STEP[layout_name], "counted (profiles/NOTES.md section 4.6)"
    return 14.2e3 + (k - 7) * (23.3e3 - 14.2e3) / 6.0, "interpolated in k between the counted k = 7 and k = 13 figures"
FLOPS_FRONT_END = 15.4e3                                                # FK, EE Jacobians, CRBA, RNEA of the Dual-UR5 tree, counted in the ISA of the
                                                                        # round-5 walk (profiles/r05_walk_isa_mix.txt; the round-4 walk did 20.1e3:
                                                                        # products with the MJCF's exact zeros and ones, a dot product per (body, hinge) pair)
MODES = {   # --dtype -> (record dtype, arithmetic label, kernel id)
    "f64": (np.float64, "f64", 0),
    "mixed": (np.float32, "f64", 3),
}


# Key order of the two dicts the driver truncates at 24 entries: what the round claims travels first
CONFIG_FIRST = ["workload", "instances_per_gpu", "layout", "k", "kernel", "rccl_ranks", "sustained_value", "from_q_value",
                "from_q_roofline_frac_fp64_valu", "from_q_parity_max_rel_err", "parity_n_checked", "parity_max_rel_err", "parity_n_over_tol",
                "parity_n_outside_domain", "eigen_path_frac", "truncated_frac", "giveups_to_generic_kernel", "secondary_mixed_value",
                "secondary_mixed_roofline_frac", "synthetic_dense_value", "end_to_end_host_arrays_value", "end_to_end_pcie_GBps",
                "end_to_end_tick_b1_us", "from_q_kernel"]
ROOFLINE_FIRST = ["bound", "achieved", "peak", "unit", "frac", "traffic", "frac_rocprof", "rocprof_avg_us", "rocprof_pass_avg_us",
                  "untraced_kernel_span_us", "untraced_period_us", "frac_from_kernel_span", "frac_from_period", "sclk_mhz", "kernel_ms",
                  "step_ms_events", "steps_per_launch", "whole_step_achieved", "algorithmic_bytes_per_launch",
                  "algorithmic_bytes_per_step_per_instance", "untraced_trains", "traffic_source", "rocprof_source", "kernel"]


def ordered_first(d, first):
    """The same dict with the keys of `first` (those present) leading, in that order."""
    out = {k: d[k] for k in first if k in d}
    out.update((k, v) for k, v in d.items() if k not in out)
    return out


def algorithmic_bytes(n, k, ndev, admittance, esz):
    """s*(n^2 + k*n + 2n + 14*ndev + [6*ndev] + n_out), n_out = n."""
    return esz * (n * n + k * n + 2 * n + 14 * ndev + (6 * ndev if admittance else 0) + n)


def measured_profile(kernel_name):
    """What the committed rocprofv3 passes say about the dominant kernel (profiles/hbm_traffic.json): HBM bytes per launch
    from the PMC passes (FETCH_SIZE x2-corrected + WRITE_SIZE) and the kernel trace's average
    duration; {} when there is no entry."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel_name, {}) or {}
    except (OSError, ValueError):
        return {}


def baseline_config_of(layout, batch, world, mode):
    if layout == "k12_admit" and batch == 65536:
        return "BASELINE configs[4]"
    if layout == "k13" and batch == 65536:
        return "BASELINE configs[2]" + ("" if mode != "f64" else " at the reference's float64")
    if layout == "k13" and batch == 32768 and world == 8:
        return "BASELINE configs[3]"
    if layout == "k13" and batch == 4096 and mode == "f64":
        return "BASELINE configs[1]"
    return "no BASELINE config"


# ---- CPU baseline: the oracle (NumPy restatement of the reference) on this box's host cores ----------------------
_CPU = {}


def effective_cores():
    """-> (usable cores, affinity count, cgroup quota in cores or None).  os.cpu_count() ignores both the affinity mask
    and a container's CPU quota; a pool of cpu_count() workers on a quota of 16 cores measures the scheduler."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2
            q, p = f.read().split()[:2]
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:                                                            # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    n = aff if quota is None else max(1, min(aff, int(quota)))
    return n, aff, quota


def _cpu_worker(wid, nworkers, seconds, barrier, q):
    """One worker = one core: the op-for-op oracle over its share of instance ids (BLAS single-threaded by the
    environment set at the top of this file).  Warm-up, then everybody starts the measured window together."""
    from oracle import osc_oracle
    a, od, gains = _CPU["a"], _CPU["od"], _CPU["gains"]
    Btot = a["M"].shape[0]
    chunk = 64
    pos = (wid * (Btot // nworkers)) % Btot

    def run(p):
        idx = [(p + i) % Btot for i in range(chunk)]
        osc_oracle.generate_batch(od, gains, a["M"], a["J"], a["dq"], a["bias"], a["ee_pose"], a["tgt_pose"],
                                  a.get("wrench"), a.get("tgt_vel"), idx=idx)
    try:
        run(pos)
        barrier.wait(timeout=300)
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            pos = (pos + chunk) % Btot
            run(pos)
            done += chunk
        q.put((wid, done, time.perf_counter() - t0))
    except Exception as e:                                  # noqa: BLE001
        q.put((wid, 0, 0.0, str(e)))


def _rb_worker(args):
    """Chained oracles for the from_q parity leg: rigid-body oracle -> records -> OSC oracle, one robot at a time."""
    lo, hi = args
    from oracle import osc_oracle, rigid_body as rb
    from irl_control_amd.rigid_body import DUAL_UR5_EE
    c = _CPU
    om = rb.Model()
    recs = [rb.records(om, c["od"], DUAL_UR5_EE, c["qpos"][b], c["qvel"][b]) for b in range(lo, hi)]
    st = {k: np.array([r[k] for r in recs]) for k in ("M", "J", "dq", "bias", "ee_pose")}
    return osc_oracle.generate_batch(c["od"], c["gains"], st["M"], st["J"], st["dq"], st["bias"], st["ee_pose"],
                                     c["tgt_pose"][lo:hi], None, None)


def from_q_reference(lay, gains, qpos, qvel, tgt_pose, nrobots, ncores):
    """Reference torques of the path from joint coordinates for the first `nrobots` robots (forked workers: the
    rigid-body oracle is ~13 ms of NumPy per robot)."""
    n = int(min(nrobots, qpos.shape[0]))
    w = max(1, min(ncores, 64, n // 16))
    _CPU.update(od=lay.as_oracle_dict(), gains=gains, qpos=qpos, qvel=qvel, tgt_pose=np.asarray(tgt_pose, dtype=np.float64))
    try:
        spans = [(n * i // (4 * w), n * (i + 1) // (4 * w)) for i in range(4 * w)]
        with mp.get_context("fork").Pool(w) as pool:
            parts = pool.map(_rb_worker, [sp for sp in spans if sp[1] > sp[0]])
        return np.concatenate(parts, axis=0)
    finally:
        _CPU.clear()


def cpu_baseline(lay, gains, arrays, seconds_single=8.0, seconds_all=8.0):
    """-> (cpu_baseline dict, reference outputs, their instance ids).  Three figures: all usable
    host cores (one oracle worker per core; affinity mask and cgroup quota honoured, all three counts stated) = the
    headline; one core; and the stacked-LAPACK restatement as a stronger single-process comparator."""
    from oracle import osc_oracle
    a64 = {k: v.astype(np.float64) for k, v in arrays.items()}
    od = lay.as_oracle_dict()

    def run(idx):
        return osc_oracle.generate_batch(od, gains, a64["M"], a64["J"], a64["dq"], a64["bias"], a64["ee_pose"],
                                         a64["tgt_pose"], a64.get("wrench"), a64.get("tgt_vel"), idx=idx)
    run(range(0, 64))                                   # warm-up
    t0 = time.perf_counter(); run(range(64, 576)); dt = time.perf_counter() - t0
    nsamp = int(max(512, min(a64["M"].shape[0] - 576, seconds_single / (dt / 512))))
    t0 = time.perf_counter(); ref = run(range(576, 576 + nsamp)); dt = time.perf_counter() - t0
    blas = os.environ.get("OPENBLAS_NUM_THREADS", "?")
    single = dict(value=nsamp / dt, unit="steps/s", cores=1,
                  sample=f"{nsamp} of the {a64['M'].shape[0]} instances of slot 0, oracle/osc_oracle.py "
                         f"(float64 NumPy, OPENBLAS_NUM_THREADS={blas}), {dt:.1f} s")
    out = dict(single)
    out["kind"] = "port"
    # whole host: one worker per usable core, forked (the arrays are inherited, not pickled); the window starts at a
    # barrier all workers reach after their warm-up, so process start-up is not in it
    ncores, aff, quota = effective_cores()
    try:
        _CPU.update(a=a64, od=od, gains=gains)
        ctx = mp.get_context("fork")
        barrier, q = ctx.Barrier(ncores), ctx.Queue()
        procs = [ctx.Process(target=_cpu_worker, args=(w, ncores, seconds_all, barrier, q), daemon=True) for w in range(ncores)]
        for p in procs:
            p.start()
        res = [q.get(timeout=600) for _ in procs]
        for p in procs:
            p.join(timeout=60)
        errs = [r[3] for r in res if len(r) > 3]
        if errs:
            raise RuntimeError(errs[0])
        total, window = sum(r[1] for r in res), max(r[2] for r in res)
        rates = sorted(r[1] / r[2] for r in res)
        value = total / window
        eff = value / (single["value"] * ncores)
        out = dict(value=value, unit="steps/s", cores=ncores, kind="port",
                   sample=f"{total} oracle steps by {ncores} worker processes in a {window:.1f} s window that starts after "
                          f"every worker's warm-up (cores: os.cpu_count() = {os.cpu_count()}, affinity mask = {aff}, cgroup CPU "
                          f"quota = {'none' if quota is None else '%.1f' % quota} -> {ncores} workers, OPENBLAS_NUM_THREADS={blas} "
                          f"each); per worker {rates[0]:.0f} / {rates[len(rates) // 2]:.0f} / {rates[-1]:.0f} steps/s (min / median "
                          f"/ max) against {single['value']:.0f} for one worker alone: parallel efficiency {eff:.2f}"
                          + ("" if eff >= 0.5 else " (the workers share SMT siblings, L3 and memory bandwidth: every step "
                             "streams its instance's 8.5 KB through NumPy temporaries)") + "; instances of slot 0",
                   parallel_efficiency=eff, single_core=single)
    except Exception as e:                                  # the baseline must never break the bench line
        out["all_cores_error"] = str(e)
    finally:
        _CPU.clear()
    # the same arithmetic with the batch axis inside NumPy's stacked LAPACK calls (no per-instance interpreter overhead)
    try:
        from oracle import osc_oracle_batched
        nb = int(min(a64["M"].shape[0], 16384))
        sl = {k: (v[:nb] if isinstance(v, np.ndarray) else v) for k, v in a64.items()}
        gb = {k: (np.asarray(v)[:nb] if np.ndim(v) > {"kp": 1, "kv": 1, "ko": 1, "k": 2, "d": 2, "max_vel": 2,
                                                       "null_kv": 0}.get(k, 99) else v) for k, v in gains.items()}
        t0 = time.perf_counter()
        osc_oracle_batched.generate_batch(od, gb, sl["M"], sl["J"], sl["dq"], sl["bias"], sl["ee_pose"], sl["tgt_pose"],
                                          sl.get("wrench"), sl.get("tgt_vel"))
        dtb = time.perf_counter() - t0
        out["vectorised"] = dict(value=nb / dtb, unit="steps/s", cores=f"one process, OPENBLAS_NUM_THREADS={blas}",
                                 sample=f"{nb} instances, oracle/osc_oracle_batched.py (stacked np.linalg calls), {dtb:.1f} s")
    except Exception as e:
        out["vectorised"] = dict(error=str(e))
    return out, ref, range(576, 576 + nsamp)


def oracle_reference(lay, gains, arrays, lo, hi):
    """float64 oracle on the (record-dtype-rounded) inputs of instances [lo, hi) -> u[hi - lo, n]."""
    from oracle import osc_oracle
    a64 = {k: v[lo:hi].astype(np.float64) for k, v in arrays.items()}
    return osc_oracle.generate_batch(lay.as_oracle_dict(), gains, a64["M"], a64["J"], a64["dq"], a64["bias"], a64["ee_pose"],
                                     a64["tgt_pose"], a64.get("wrench"), a64.get("tgt_vel"))


def parity_sample(arr, u, ref, idx, tol=1e-5, note=None):
    """GPU vs float64 oracle on the same (record-dtype-rounded) inputs: u[i] against ref[i] for i in idx; the instances
    over `tol` are then classified against the parity domain (the reference's own answer well
    defined, no singular value within 1 % of the pinv cut)."""
    from oracle import osc_oracle
    idx = np.asarray(list(idx))
    err = np.max(np.abs(u[idx].astype(np.float64) - ref[idx]), axis=1) / np.max(np.abs(ref[idx]), axis=1)
    over = idx[err > tol]

    def in_domain(b):
        Mx, Minv, Mxi, det = osc_oracle.task_inertia(arr["J"][b].astype(np.float64), arr["M"][b].astype(np.float64))
        sv = np.linalg.svd(Mxi, compute_uv=False)
        if abs(det) >= 1e-4:
            return bool(sv[-1] > 1e-12 * sv[0])
        return not np.any(np.abs(sv / sv[0] / 1e-5 - 1.0) < 1e-2)
    in_dom = sum(in_domain(b) for b in over[:2000])
    # how many of the CHECKED instances lie outside the parity domain at all (whatever their error): on a bounded sub-sample
    exam = idx[:: max(1, len(idx) // 8192)]
    n_out = sum(not in_domain(b) for b in exam)
    return {"n": int(len(idx)), "tolerance": tol, "median_rel_err": float(np.median(err)),
            "p99_rel_err": float(np.quantile(err, 0.99)), "max_rel_err": float(err.max()), "n_over_tol": int(len(over)),
            "n_over_tol_in_parity_domain": int(in_dom),
            "n_outside_parity_domain": int(n_out), "n_examined_for_the_domain": int(len(exam)),
            "note": note or "GPU vs float64 oracle on the same (record-dtype-rounded) inputs; parity domain per SURVEY.md 8c"}


def measure_from_q(BatchedOSC, synth, args, B, local_rank, state0=None, ref_u=None):
    """The path from joint coordinates: per step the rigid-body front end computes M, J,
    bias and the EE poses from resident (qpos, qvel) on the GPU, then the OSC step runs on them.  What a host-side
    simulator would have to ship per step otherwise is the 8.5 KB of records (PCIe ceiling ~63 GB/s / 8 536 B = 7.4e6
    steps/s in float64, 1.5e7 in float32).  The bound of this path is arithmetic, not HBM (568 B in, 200 B out per
    robot): `roofline` prices it against the fp64 vector peak."""
    from irl_control_amd.rigid_body import RigidBodyModel
    try:
        dt, arith, kern = MODES[args.dtype]
        lay = synth.make_layout(args.layout)
        model = RigidBodyModel.load("dual_ur5")
        osc = BatchedOSC(lay, B, dtype=dt, hip_device=local_rank, n_slots=args.slots, kernel=kern)
        osc.set_model(model)
        rng = np.random.default_rng(20241008 + 78)
        _, gains, arr = synth.make_batch(args.layout, B, seed=20241008 + 2000, dtype=dt)
        osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
        for s in range(args.slots):
            qpos, qvel = state0 if (s == 0 and state0 is not None) else model.random_state(rng, B)
            osc.upload_q(qpos, qvel, slot=s)
            osc.frontend(slot=s)
            osc.set_targets(arr["tgt_pose"], arr.get("tgt_vel"), slot=s)
        steps = max(64, min(256, args.steps // 32 * 8))          # whole trains of 8, enough of them for a steady state
        osc.step_resident_from_q(max(24, min(args.preroll, 400)))      # untimed: the clocks settle (as in measure())
        osc.device_sync()
        t0 = time.perf_counter()
        ms_total, ms_step = osc.step_resident_from_q(steps)
        osc.device_sync()
        el = time.perf_counter() - t0
        trains = osc.time_trains(64, from_q=True) if "fused" in osc.from_q_name else None
        for s in range(args.slots):                     # a fused step leaves no records in the slot (include/irlosc.h)
            osc.frontend(slot=s)
        _, ms_osc = osc.step_resident(steps)
        esz = np.dtype(dt).itemsize
        f_osc, f_how = flops_osc_step(args.layout, lay.k)
        flops = FLOPS_FRONT_END + f_osc
        achieved = flops * B / (ms_step * 1e-3) / 1e12
        res = dict(value=B * steps / el, unit="steps/s", ms_per_step=el / steps * 1e3, ms_per_step_events=ms_step,
                   ms_osc_step_on_records_alone=ms_osc, kernel=osc.from_q_name,
                   input_bytes_per_step_per_instance=2 * lay.n * 8 + 7 * lay.ndev * esz,
                   hbm_traffic_bytes_per_step_per_instance=_fromq_traffic(osc.from_q_name, B),
                   roofline=dict(bound="fp64_valu", achieved=achieved, peak=FP64_VALU_PEAK_TFLOPS, unit="TFLOP/s",
                                 frac=achieved / FP64_VALU_PEAK_TFLOPS, flops_per_step_per_instance=flops, flops_osc_step=f_how,
                                 note="useful fp64 flops (FMA = 2) of front end + OSC step, profiles/NOTES.md section 4.6; HBM sees "
                                      "568 B in and 200 B out per robot, so the HBM roof is not the bound of this path"),
                   untraced=train_summary(trains, osc.steps_per_launch) if trains is not None else None,
                   note="per step: rigid-body front end (FK, EE Jacobians, CRBA, RNEA) from resident (qpos, qvel) and the OSC step on "
                        "what it leaves behind (fused path: a compact exchange buffer of the structural non-zeros, 2.6 KB per robot, "
                        "instead of 8.5 KB of dense records; round 6: the OSC step too runs one lane per robot on that buffer, the ~15 % of "
                        "robots whose solve is a truncated pseudo-inverse finish in an eigen pass, again one lane per robot); nothing crosses PCIe")
        if ref_u is not None:                          # front end + step on slot 0 against the chained oracles
            u = osc.step_q(slot=0)
            res["parity_sample"] = _parity_plain(u, ref_u, 1e-5,
                                                 "GPU front end + OSC step vs oracle/rigid_body.py -> oracle/osc_oracle.py chained "
                                                 "on the same (qpos, qvel, targets) of the first robots of slot 0")
        osc.close()
        return res
    except Exception as e:                              # never break the bench line
        return dict(error=str(e))


def _fromq_traffic(name, B):
    """HBM bytes per robot and step of the fused path from the committed PMC passes (profiles/hbm_traffic.json: every kernel of the
    path, FETCH_SIZE x2 + WRITE_SIZE; an upper bound for the gathered 8-byte loads), or None."""
    if "fused" not in name:
        return None
    keys = (["fromq_lane:walk", "fromq_lane:osc_lane", "fromq_lane:eigen_pass"] if "osc_lane" in name
            else ["osc_frontend_lane_compact_dual_ur5", "osc_row16_f64_n25_k13_fromq"])
    ents = [measured_profile(k) for k in keys]
    if not all(e and e.get("instances") == B and e.get("steps_per_launch") == ents[0].get("steps_per_launch") for e in ents):
        return None
    return sum(e["traffic_bytes_per_launch"] for e in ents) / (ents[0]["steps_per_launch"] * B)


def _parity_plain(u, ref, tol, note):
    n = len(ref)
    err = np.max(np.abs(u[:n].astype(np.float64) - ref), axis=1) / np.max(np.abs(ref), axis=1)
    return {"n": int(n), "tolerance": tol, "median_rel_err": float(np.median(err)), "p99_rel_err": float(np.quantile(err, 0.99)),
            "max_rel_err": float(err.max()), "n_over_tol": int((err > tol).sum()), "note": note}


def train_summary(tt, spl):
    """irlosc_time_trains -> the three figures that reconcile a kernel trace with the driver's clock: the in-kernel duration of a
    train (first wave's start to last wave's end, the kernel's own 100 MHz clock: what a trace reports per dispatch without the
    tracer), the steady-state PERIOD between consecutive trains' starts (what ms_per_step x steps_per_launch measures), and the
    HIP event pair around a train."""
    tt = np.asarray(tt)
    dur = tt[:, 2] - tt[:, 1]
    per = np.diff(tt[:, 1])
    q = lambda a, x: float(np.quantile(a, x))
    return {"trains": int(len(tt)), "steps_per_train": int(spl),
            "kernel_span_us": {"median": q(dur, 0.5), "p10": q(dur, 0.1), "p90": q(dur, 0.9)},
            "period_us": {"median": q(per, 0.5), "p10": q(per, 0.1), "p90": q(per, 0.9), "mean": float(per.mean())},
            "event_pair_us": {"median": q(tt[:, 0], 0.5) * 1e3, "mean": float(tt[:, 0].mean()) * 1e3},
            "overlap_us_median": q(dur[:-1] - per, 0.5),
            "sclk_mhz": ({"median": q(tt[:, 3], 0.5), "p10": q(tt[:, 3], 0.1), "p90": q(tt[:, 3], 0.9)} if tt.shape[1] > 3 and tt[:, 3].max() > 0 else None),
            "note": "untraced, one run: kernel_span = last wave's end - first wave's start of a train's main kernel (s_memrealtime stamped "
                    "in the kernel); period = start-to-start of consecutive trains; overlap = span - period (> 0: the next train's "
                    "first waves run while this train's last waves drain -- kernels of one stream are not serialised by a barrier); sclk_mhz = "
                    "shader cycle counter against wall clock over one sample wave per train: the sustained clock under this fp64 load (peak 2 400)"}


def measure_end_to_end(BatchedOSC, lay, gains, arr, dt, kern, local_rank):
    """SURVEY.md 8(d) "separately end-to-end with H2D/D2H": the boundary handing over HOST arrays every tick.  Never `value`."""
    try:
        B = arr["M"].shape[0]
        esz = np.dtype(dt).itemsize
        osc = BatchedOSC(lay, B, dtype=dt, hip_device=local_rank, kernel=kern)
        osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
        a = (arr["M"], arr["J"], arr["dq"], arr["bias"], arr["ee_pose"], arr["tgt_pose"], arr.get("tgt_vel"), arr.get("wrench"))
        for _ in range(2):
            osc.generate_batched(*a)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); osc.generate_batched(*a); ts.append(time.perf_counter() - t0)
        el = float(np.median(ts))
        nbytes = sum(x.nbytes for x in a if x is not None) + B * lay.n * esz + 4 * B
        out = {"generate_batched": {"value": B / el, "unit": "steps/s", "ms_per_tick": el * 1e3, "instances": B,
                                    "pcie_GBps": nbytes / el / 1e9, "bytes_per_instance": nbytes // B, "kernel": osc.kernel_name,
                                    "what": "BatchedOSC.generate_batched from pageable host arrays: irlosc_upload (records + device-side symmetry / "
                                            "structure probes) + irlosc_set_targets + irlosc_step with the torques and flags copied back"}}
        osc.close()
        if esz == 8:      # the same boundary with float32 records (fp64 arithmetic: the mixed path): half the bytes over the link
            a32 = tuple(None if x is None else np.ascontiguousarray(x, dtype=np.float32) for x in a)
            osc = BatchedOSC(lay, B, dtype=np.float32, hip_device=local_rank)
            osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
            for _ in range(2):
                osc.generate_batched(*a32)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); osc.generate_batched(*a32); ts.append(time.perf_counter() - t0)
            el32 = float(np.median(ts))
            nb32 = sum(x.nbytes for x in a32 if x is not None) + B * lay.n * 4 + 4 * B
            out["generate_batched_float32"] = {"value": B / el32, "unit": "steps/s", "ms_per_tick": el32 * 1e3, "instances": B,
                                               "pcie_GBps": nb32 / el32 / 1e9, "bytes_per_instance": nb32 // B, "kernel": osc.kernel_name,
                                               "what": "generate_batched from float32 host arrays (float32 records, fp64 arithmetic)"}
            osc.close()
        # B = 1: what OSC.generate costs per tick (irlosc_tick: pack, one H2D, the step, one D2H, one synchronisation)
        one = {k: (v[:1] if isinstance(v, np.ndarray) else v) for k, v in arr.items()}
        osc = BatchedOSC(lay, 1, dtype=dt, hip_device=local_rank, kernel=kern)
        osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
        a1 = (one["M"], one["J"], one["dq"], one["bias"], one["ee_pose"], one["tgt_pose"], one.get("tgt_vel"), one.get("wrench"))
        for _ in range(50):
            osc.tick(*a1)
        t1 = []
        for _ in range(300):
            t0 = time.perf_counter(); osc.tick(*a1); t1.append(time.perf_counter() - t0)
        out["tick_b1_us"] = {"median": float(np.median(t1)) * 1e6, "p99": float(np.quantile(t1, 0.99)) * 1e6, "kernel": osc.kernel_name,
                             "what": "irlosc_tick at B = 1 (the call under OSC.generate)"}
        osc.close()
        # raw simulator arrays in, state assembly on the GPU (irlosc_upload_raw), then the step: k13 only (the mapping below)
        if lay.k == 13 and lay.ndev == 3:
            from irl_control_amd import _lib
            nv, ns = 25, 18
            d = _lib.RawDesc()
            d.nv, d.n_sensor = nv, ns
            for p_ in range(32):
                d.joint_ids[p_] = p_ if p_ < 25 else 0
                d.dq_src[p_] = p_ if p_ < 25 else -1
            for i in range(4):
                d.ft_force0[i], d.ft_torque0[i] = -1, -1
            jacp, jacr = np.zeros((B, 3, 3, nv), dt), np.zeros((B, 3, 3, nv), dt)
            jacp[:, 0], jacr[:, 0] = arr["J"][:, 0:3], arr["J"][:, 3:6]
            jacp[:, 1], jacr[:, 1] = arr["J"][:, 6:9], arr["J"][:, 9:12]
            jacr[:, 2, 2] = arr["J"][:, 12]
            raw = dict(qM=arr["M"], qvel=arr["dq"], qfrc_bias=arr["bias"], jacp=jacp, jacr=jacr,
                       ee_xpos=np.ascontiguousarray(arr["ee_pose"][:, :, :3]), ee_xquat=np.ascontiguousarray(arr["ee_pose"][:, :, 3:]))
            osc = BatchedOSC(lay, B, dtype=dt, hip_device=local_rank, kernel=kern)
            osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
            ts = []
            for it in range(5):
                t0 = time.perf_counter()
                osc.upload_raw(d, **raw)
                osc.set_targets(arr["tgt_pose"])
                osc.step()
                ts.append(time.perf_counter() - t0)
            el = float(np.median(ts[1:]))
            rb = sum(x.nbytes for x in raw.values()) + arr["tgt_pose"].nbytes + B * lay.n * esz + 4 * B
            out["upload_raw_step"] = {"value": B / el, "unit": "steps/s", "ms_per_tick": el * 1e3, "pcie_GBps": rb / el / 1e9,
                                      "what": "raw simulator arrays (mj_fullM, jacp / jacr, qvel, qfrc_bias, xpos / xquat; nv = 25) from the host, "
                                              "state assembly on the GPU (irlosc_upload_raw), targets, one step, torques back"}
            # the same with M as MuJoCo holds it (mjData.qM, nM = 155 entries per robot: irlosc_upload_raw_sparse expands it on the GPU --
            # robot.py:68-72 has mj_fullM do that on the host), float64 and float32 arrays: the best case of the host-fed deployment
            from irl_control_amd import raw as rawmod
            PARENT = [-1, 0, 1, 2, 3, 4, 5, 6, 7, 6, 6, 10, 6, 0, 13, 14, 15, 16, 17, 18, 19, 18, 18, 22, 18]      # dof_parentid of the Dual-UR5
            ql = rawmod.qm_layout(PARENT)
            tree_ok = bool(np.all(rawmod.pack_qM(arr["M"][:64], ql).sum() != 0)) and bool(osc.slot_structure(0))
            osc.close()
            if tree_ok:                               # (records with the tree's zeros: the physical workload; a synthetic dense M has no sparse form)
                for tag, dts in (("upload_raw_sparse_step", dt), ("upload_raw_sparse_step_float32", np.float32)):
                    rs = {k: np.ascontiguousarray(v, dtype=dts) for k, v in raw.items() if k != "qM"}
                    qs = np.ascontiguousarray(rawmod.pack_qM(arr["M"], ql), dtype=dts)
                    tg = np.ascontiguousarray(arr["tgt_pose"], dtype=dts)
                    osc = BatchedOSC(lay, B, dtype=dts, hip_device=local_rank)
                    osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
                    ts = []
                    for it in range(5):
                        t0 = time.perf_counter()
                        osc.upload_raw(d, qs, qm_layout=ql, **rs)
                        osc.set_targets(tg)
                        osc.step()
                        ts.append(time.perf_counter() - t0)
                    el = float(np.median(ts[1:]))
                    es2 = np.dtype(dts).itemsize
                    rb = qs.nbytes + sum(x.nbytes for x in rs.values()) + tg.nbytes + B * lay.n * es2 + 4 * B
                    out[tag] = {"value": B / el, "unit": "steps/s", "ms_per_tick": el * 1e3, "pcie_GBps": rb / el / 1e9, "bytes_per_instance": rb // B,
                                "kernel": osc.kernel_name,
                                "what": "as upload_raw_step, with mjData.qM in MuJoCo's sparse form (nM = 155 values per robot) expanded on the GPU"}
                    osc.close()
        out["note"] = ("host arrays cross PCIe every tick: the link bounds these figures, not the kernel; `value` (inputs resident in HBM) "
                       "is the headline, and from_q (568 B per robot in) is the path that needs no dense records at all")
        return out
    except Exception as e:                              # never break the bench line
        return dict(error=str(e))


def self_launch(args_list, n):
    """`python bench.py --gpus N` without a launcher environment: start the N workers here, one per GPU, with the
    variables a launcher would provide; rank 0's stdout (the JSON line) passes through.  -> exit code."""
    import secrets
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tag = f"self_{port}_{secrets.token_hex(4)}"
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), IRLOSC_RDV_TAG=tag)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + args_list, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    try:
        from irl_control_amd import sharding
        d = sharding.rendezvous_dir(tag)
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)
    except OSError:
        pass
    return next((rc for rc in rcs if rc), 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--preroll", type=int, default=1000,
                    help="extra UNTIMED steps before the warm-up: after the idle set-up phase the power management needs ~50 ms of load to settle")
    ap.add_argument("--dtype", default="f64", choices=sorted(MODES))
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--total-batch", type=int, default=0, help="instances over all GPUs (overrides --batch)")
    ap.add_argument("--slots", type=int, default=16,
                    help="resident input batches the steps rotate over.  Consecutive trains of 8 steps run on two streams, so two trains "
                         "are in flight: with 16 slots they never read the same batch (with fewer, the second reader of a batch is served "
                         "from the Infinity Cache and the rate is inflated: profiles/NOTES.md, round 6)")
    ap.add_argument("--layout", default="k13")
    ap.add_argument("--kernel", type=int, default=-1, help="override: 0 auto, 1 generic, 3 row16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="length of each of the two CPU timing windows (one core, all cores)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other storage variant (mixed) and the synthetic-records run")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the H2D/D2H-inclusive legs (generate_batched from host arrays, tick at B = 1, upload_raw)")
    ap.add_argument("--require-rccl", action="store_true", help="N > 1: exit non-zero unless every rank's barrier / reduction went through RCCL")
    ap.add_argument("--sustained-steps", type=int, default=100000,
                    help="steps of the long-run leg reported as `sustained` (0: skip); ~10 s of back-to-back trains: long enough for the driver's GPU sampler")
    ap.add_argument("--slices", type=int, default=1, help="sub-batches ('virtual ranks') per rank, one output checksum each")
    ap.add_argument("--no-from-q", action="store_true", help="skip the joint-coordinates path (front end + step)")
    ap.add_argument("--workload", default="physical", choices=["physical", "synthetic"],
                    help="physical: records of random joint states from the front end (tree zeros); synthetic: random dense SPD M")
    ap.add_argument("--mint-physical", default=None, help=argparse.SUPPRESS)   # internal: write slot 0 of the physical workload to this directory
    ap.add_argument("--as-rank", type=int, default=-1, help=argparse.SUPPRESS)  # internal: seed the data like this rank (the minting child of rank 0)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    from irl_control_amd import BatchedOSC, sharding, synth
    rank, world, local_rank = sharding.env_world()
    if args.as_rank >= 0:
        rank = args.as_rank
    if "IRLOSC_BENCH_DEVICE" in os.environ:                  # test hook: several ranks on ONE GPU (then RCCL refuses the duplicate
        local_rank = int(os.environ["IRLOSC_BENCH_DEVICE"])  # device and the file-based reduction is what gets exercised)
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    B = args.total_batch // world if args.total_batch else args.batch
    S = max(1, args.slices)
    if B % S:
        sys.exit(f"--slices {S} does not divide the {B} instances of a rank")
    Bs = B // S
    comm, comm_note = None, "single process"

    def vrank(j):                        # "virtual rank" of sub-batch j of this rank: what seeds its data
        return rank * S + j

    def make_slot(mode, s):
        parts = [synth.make_batch(args.layout, Bs, seed=20241008 + 1000 * 2 + 17 * s + 101 * vrank(j), dtype=MODES[mode][0]) for j in range(S)]
        lay_, gains_ = parts[0][0], parts[0][1]
        arr_ = {k: np.concatenate([p_[2][k] for p_ in parts], axis=0) for k in parts[0][2]} if S > 1 else parts[0][2]
        return lay_, gains_, arr_

    def fill_physical(osc, lay, dt, s, model):
        """Slot s of the physical workload: random joint states -> front end (records stay in HBM) -> targets around the end
        effectors it found (-> wrench for the admittance layout, which only arrives with an upload of records).  Returns
        slot 0's records as host arrays (what the oracle legs are run on), None for the other slots."""
        rngs = [np.random.default_rng(20241008 + 1000 * 2 + 17 * s + 101 * vrank(j) + 500000) for j in range(S)]
        st = [model.random_state(r_, Bs) for r_ in rngs]
        qpos, qvel = np.concatenate([x_[0] for x_ in st]), np.concatenate([x_[1] for x_ in st])
        osc.upload_q(qpos, qvel, slot=s)
        osc.frontend(slot=s)
        ee = osc.download_records(s, keys=("ee_pose",))["ee_pose"]
        tgt = np.concatenate([synth.targets_near(ee[j * Bs:(j + 1) * Bs].astype(np.float64), rngs[j]) for j in range(S)]).astype(dt)
        wrench = (np.concatenate([rngs[j].normal(0.0, 5.0, size=(Bs, lay.ndev, 6)) for j in range(S)]).astype(dt)
                  if lay.admittance else None)
        rec = None
        if wrench is not None or s == 0:
            rec = osc.download_records(s)
        if wrench is not None:
            osc.upload(rec["M"], rec["J"], rec["dq"], rec["bias"], rec["ee_pose"], wrench, slot=s)
        osc.set_targets(tgt, None, slot=s)
        if s != 0:
            return None
        rec["tgt_pose"] = tgt
        if wrench is not None:
            rec["wrench"] = wrench
        return rec

    def fill_slots(osc, lay, mode, workload):
        """-> (gains, slot 0 as host arrays)."""
        dt = MODES[mode][0]
        slot0 = None
        if workload == "physical":
            from irl_control_amd.rigid_body import RigidBodyModel
            model = RigidBodyModel.load("dual_ur5")
            osc.set_model(model)
            _, gains, _ = synth.make_batch(args.layout, 2, seed=1, dtype=dt)       # the YAML gain set of the layout
            osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
            for s in range(args.slots):
                r = fill_physical(osc, lay, dt, s, model)
                slot0 = r if s == 0 else slot0
            return gains, slot0
        for s in range(args.slots):
            _, gains, arr = make_slot(mode, s)
            osc.upload(arr["M"], arr["J"], arr["dq"], arr["bias"], arr["ee_pose"], arr.get("wrench"), slot=s)
            osc.set_targets(arr["tgt_pose"], arr.get("tgt_vel"), slot=s)
            if s == 0:
                slot0 = arr
                osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"],
                              gains["null_kv"])
            else:
                del arr
        return gains, slot0

    if args.mint_physical:        # internal: slot 0 of the physical workload as .npy files, for the CPU legs of the parent process
        dt = MODES[args.dtype][0]
        lay = synth.make_layout(args.layout)
        osc = BatchedOSC(lay, B, dtype=dt, hip_device=local_rank, n_slots=args.slots, kernel=MODES[args.dtype][2])
        from irl_control_amd.rigid_body import RigidBodyModel
        model = RigidBodyModel.load("dual_ur5")
        osc.set_model(model)
        rec = fill_physical(osc, lay, dt, 0, model)
        osc.close()
        for k2, v in rec.items():
            np.save(os.path.join(args.mint_physical, k2 + ".npy"), v)
        return

    def measure(mode, steps, warmup, preroll, workload, long_leg=True):
        dt, arith, kern = MODES[mode]
        if args.kernel >= 0:
            kern = args.kernel
        esz = np.dtype(dt).itemsize
        lay = synth.make_layout(args.layout)
        osc = BatchedOSC(lay, B, dtype=dt, hip_device=local_rank, n_slots=args.slots, kernel=kern)
        gains, slot0 = fill_slots(osc, lay, mode, workload)
        if preroll > 0:                                          # untimed: lets the clocks settle after the idle set-up
            osc.step_resident(preroll)
        if warmup > 0:
            osc.step_resident(warmup)
        # the timed region: EXACTLY `steps` steps between barrier + device synchronisation on both sides
        if comm:
            comm.barrier()
        osc.device_sync()
        t0 = time.perf_counter()
        ms_total, ms_kernel = osc.step_resident(steps)       # HIP events on the library's own stream ride along
        osc.device_sync()
        if comm:
            comm.barrier()
        elapsed = time.perf_counter() - t0
        total_steps, elapsed, rate = sharding.reduce_throughput(B * steps, elapsed, comm)
        # the same bracket over a LONG run (the driver's command times 20 steps = 2 ms; this is the figure that does not depend on
        # how the first milliseconds went): not `value`, reported beside it
        sustained = None
        if args.sustained_steps > 0 and long_leg:
            if comm:
                comm.barrier()
            osc.device_sync()
            t1 = time.perf_counter()
            osc.step_resident(args.sustained_steps)
            osc.device_sync()
            if comm:
                comm.barrier()
            el1 = time.perf_counter() - t1
            _, el1, rate1 = sharding.reduce_throughput(B * args.sustained_steps, el1, comm)
            sustained = {"steps": args.sustained_steps, "value": rate1, "unit": "steps/s", "ms_per_step": el1 / args.sustained_steps * 1e3,
                         "seconds": el1}
        bytes_step = algorithmic_bytes(lay.n, lay.k, lay.ndev, lay.admittance, esz) * B
        spl = osc.steps_per_launch              # the throughput paths chain this many steps into one launch
        bytes_launch = bytes_step * spl
        # dominant kernel alone, HIP events on its stream (one event pair per launch of a train)
        ms_dom = osc.time_dominant_kernel(min(200, max(10 * spl, steps // 2)))
        if comm:
            ms_dom = comm.reduce(0.0, ms_dom)[1]
            ms_kernel = comm.reduce(0.0, ms_kernel)[1]
        achieved = bytes_launch / (ms_dom * 1e-3) / 1e9
        tree = all(osc.slot_structure(sl) for sl in range(args.slots))
        kname = osc.kernel_name + ("+tree" if tree else "")
        note = ""
        if "row16" in kname:
            tpass = os.environ.get("IRLOSC_TASK_PASS", "1") != "0"
            note = ":step(" + ("the task pass osc_task_rows_dense_kernel (part 1 of the task signal, one lane per (instance, device)) + the row16 "
                               "kernel: " if tpass else "") + "all instances incl. the in-kernel truncated-pinv stage; + the give-up list launch)" + (
                "; tree-structured factorisation M = L^T L on the dense records (their zero pattern verified at upload / by construction)"
                if tree else "")
        prof = measured_profile(kname)
        same = prof.get("instances", B) == B and prof.get("steps_per_launch", spl) == spl
        # `traffic` is a PMC figure: it cannot be taken in an untraced run, so the line carries the one of the committed passes
        # (profiles/hbm_traffic.json) and says so under committed_profile; everything else in `roofline` is measured in THIS run
        roof = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS,
                    # (PMC bytes of the step's kernels: the row16 kernel + the task pass ahead of it)
                    traffic=(prof.get("traffic_bytes_per_launch") + (prof.get("pass_traffic_bytes_per_launch") or 0.0))
                    if same and prof.get("traffic_bytes_per_launch") else None,
                    traffic_source="committed_profile (PMC passes need a tracer; not measured in this run)" if same and prof.get("traffic_bytes_per_launch") else None,
                    kernel=kname + note, kernel_ms=ms_dom, steps_per_launch=spl, step_ms_events=ms_kernel,
                    whole_step_achieved=bytes_step / (ms_kernel * 1e-3) / 1e9,
                    algorithmic_bytes_per_launch=bytes_launch,
                    algorithmic_bytes_per_step_per_instance=bytes_step // B)
        if "row16" in kname:
            # untraced evidence of the same run: per-train event pairs + the wall clock stamped inside the kernel
            ntr = 256 if steps >= 1000 else 32
            roof["untraced"] = train_summary(osc.time_trains(ntr), spl)
            roof["untraced"]["frac_from_period"] = bytes_launch / (roof["untraced"]["period_us"]["median"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            roof["untraced"]["frac_from_kernel_span"] = bytes_launch / (roof["untraced"]["kernel_span_us"]["median"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            # the same figures as FLAT scalars: the driver's record of the line keeps the scalars of `roofline` / `config` only
            ut = roof["untraced"]
            roof.update(untraced_kernel_span_us=ut["kernel_span_us"]["median"], untraced_period_us=ut["period_us"]["median"],
                        untraced_trains=ut["trains"], sclk_mhz=(ut["sclk_mhz"] or {}).get("median"),
                        frac_from_kernel_span=ut["frac_from_kernel_span"], frac_from_period=ut["frac_from_period"])
        if prof.get("rocprof_avg_us") and same:
            # the COMMITTED rocprofv3 passes of this command (profiles/; not measured now): kernel trace average and PMC traffic
            roof["committed_profile"] = {
                "file": "profiles/hbm_traffic.json", "entry": kname,
                "rocprof_avg_us": prof["rocprof_avg_us"], "traffic_bytes_per_launch": prof.get("traffic_bytes_per_launch"),
                "rocprof_pass_avg_us": prof.get("rocprof_pass_avg_us"),
                # the step = task pass + row16 kernel: both durations of the committed trace count
                "frac_rocprof": bytes_launch / ((prof["rocprof_avg_us"] + (prof.get("rocprof_pass_avg_us") or 0.0)) * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "note": "a kernel trace serialises dispatches: the row16 kernel's per-dispatch average corresponds to untraced.kernel_span_us "
                        "(that kernel alone), the sum with the task pass to ms_per_step x steps_per_launch (= untraced.period_us, which also "
                        "holds the give-up launch and the gaps); profiles/README.md"}
            roof.update(rocprof_avg_us=prof["rocprof_avg_us"], rocprof_pass_avg_us=prof.get("rocprof_pass_avg_us"),
                        frac_rocprof=roof["committed_profile"]["frac_rocprof"],
                        rocprof_source="profiles/hbm_traffic.json (committed kernel trace of this command on another box; not measured in this run)")
        res = dict(value=rate, ms_per_step=elapsed / steps * 1e3, kernel=kname, mode=mode, arith=arith,
                   records="float64" if esz == 8 else "float32", layout=lay, roofline=roof, workload=workload, sustained=sustained)
        osc.step(slot=0)
        u, fl = osc.download(B)
        res["giveups"] = int(osc.giveup_counts()[0])
        res["checksum"] = sharding.checksum_u64(u)
        res["slice_checksums"] = [sharding.checksum_u64(u[j * Bs:(j + 1) * Bs]) for j in range(S)]
        osc.close()
        return res, (lay, gains, slot0, u, fl)

    # CPU legs first: their worker processes are forked before this process has initialised the HIP runtime
    cb = ref = ref_idx = fq_state = fq_ref = None
    others = [m for m in ("f64", "mixed") if m != args.dtype] if (world == 1 and not args.no_secondary) else []
    NSEC = 4096                       # instances of slot 0 the secondary modes are checked on (oracle in this process)
    minted_crc = None
    if rank == 0 and not args.no_cpu_baseline:          # N > 1 too: rank 0's host cores, rank 0's shard (the other ranks wait in make_comm)
        if args.workload == "physical":
            # The records of the physical workload come from the GPU front end, and the CPU legs fork their workers before
            # THIS process touches the HIP runtime: a child process computes slot 0 (same seeds, same kernel: the same bits,
            # checked below) and leaves it in a private directory.
            import shutil
            import tempfile
            base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
            arr0, mint_error = None, None
            for where in (base, None):                   # /dev/shm first (0.5 GB of records), then the default temporary directory
                d = tempfile.mkdtemp(prefix="irlosc_mint_", dir=where)
                try:
                    if os.environ.get("IRLOSC_BENCH_FAIL_MINT"):          # test hook: the hand-over fails, the line must still come out
                        raise OSError("IRLOSC_BENCH_FAIL_MINT is set")
                    cmd = [sys.executable, os.path.abspath(__file__), "--mint-physical", d, "--batch", str(B), "--layout", args.layout,
                           "--dtype", args.dtype, "--slots", "1", "--slices", str(S), "--as-rank", str(rank)]
                    env = {k_: v_ for k_, v_ in os.environ.items()
                           if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "IRLOSC_RDV_TAG")}
                    env["IRLOSC_BENCH_DEVICE"] = str(local_rank)
                    subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL)
                    arr0 = {f[:-4]: np.load(os.path.join(d, f)) for f in sorted(os.listdir(d)) if f.endswith(".npy")}
                    break
                except (OSError, subprocess.CalledProcessError) as e:      # the baseline must never break the bench line
                    mint_error = str(e)
                finally:
                    shutil.rmtree(d, ignore_errors=True)
                if where is None:
                    break
            lay0 = synth.make_layout(args.layout)
            _, gains0, _ = synth.make_batch(args.layout, 2, seed=1, dtype=MODES[args.dtype][0])
            if arr0 is not None:
                minted_crc = sharding.checksum_u64(arr0["M"])
        else:
            lay0, gains0, arr0 = make_slot(args.dtype, 0)
        if arr0 is None:         # no physical slot for the CPU legs: time the oracle on the synthetic batch (same arithmetic per step)
            lay0, gains0, arr0 = make_slot(args.dtype, 0)
            cb, _, _ = cpu_baseline(lay0, gains0, arr0, args.cpu_seconds, args.cpu_seconds)
            cb["sample"] += f"; SYNTHETIC records (slot 0 of the physical workload could not be handed to the CPU legs: {mint_error})"
        else:
            cb, ref, ref_idx = cpu_baseline(lay0, gains0, arr0, args.cpu_seconds, args.cpu_seconds)
        if world == 1 and not args.no_from_q and args.layout in ("k13", "k7"):
            from irl_control_amd.rigid_body import RigidBodyModel
            model = RigidBodyModel.load("dual_ur5")
            fq_state = model.random_state(np.random.default_rng(20241008 + 77), B)
            _, gq, aq = synth.make_batch(args.layout, B, seed=20241008 + 2000, dtype=MODES[args.dtype][0])
            fq_ref = from_q_reference(lay0, gq, fq_state[0], fq_state[1], aq["tgt_pose"], 4096, effective_cores()[0])
            del aq
        del arr0
    if world > 1:        # behind the CPU legs: RCCL initialises the HIP runtime, and the CPU workers are forked before that
        comm, comm_note = sharding.make_comm(rank, world, local_rank, init_timeout_s=900.0)
        if args.require_rccl and not isinstance(comm, sharding.RcclComm):
            if rank == 0:
                print(f"bench.py: --require-rccl, but the ranks fell back: {comm_note}", file=sys.stderr, flush=True)
            comm.close()
            sys.exit(3)
    primary, chk = measure(args.dtype, args.steps, args.warmup, preroll=args.preroll, workload=args.workload)
    if minted_crc is not None and sharding.checksum_u64(chk[2]["M"]) != minted_crc:
        raise RuntimeError("slot 0 of the physical workload differs between the minting process and this one")
    lay = primary.pop("layout")
    names = ", ".join(f"{nm}:{r}" for nm, r in zip(lay.dev_names, lay.dev_rows))
    out = {
        "metric": "OSC control steps/sec", "value": primary["value"], "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": primary["ms_per_step"],
        "higher_is_better": True, "scaling": "strong" if args.total_batch else "weak", "vs_baseline": None,
        "dtype": primary["arith"],
        "data": "synthetic (" + ("records of uniformly random joint states from the rigid-body front end" if args.workload == "physical"
                                 else "random dense SPD M, SURVEY.md 8d") + "; no dataset, no checkpoint)",
        "config": {"workload": f"{B} Dual-UR5 instances per GPU ({baseline_config_of(args.layout, B, world, args.dtype)}), "
                               + ("records of uniformly random joint states (SURVEY.md 8d: true CRBA on random qpos) computed once by "
                                  "the rigid-body front end before the timed region, targets scattered around the end effectors, "
                                  if args.workload == "physical" else "synthetic records (random dense SPD M, SURVEY.md 8d), ") +
                               f"n={lay.n} joints, layout {args.layout}: k={lay.k} task rows over {lay.ndev} target devices "
                               f"({names}), gravity {'on' if lay.use_g else 'off'}, null-space {'on' if lay.nullspace else 'off'}, "
                               f"admittance wrench term {'on' if lay.admittance else 'off'}, {primary['records']} records, "
                               f"{primary['arith']} arithmetic, inputs resident in HBM, {args.slots} rotating batches",
                   "instances_per_gpu": B, "layout": args.layout, "k": lay.k, "ndev": lay.ndev,
                   "admittance": bool(lay.admittance), "records": primary["records"], "arithmetic": primary["arith"],
                   "kernel": primary["kernel"], "records_from": args.workload, "preroll_steps": args.preroll,
                   "steps_per_launch": primary["roofline"]["steps_per_launch"],
                   "sharding": f"{world} x independent shards, no data-path collective; barrier and final sum(steps) / "
                               f"max(elapsed) reduction: {comm_note}",
                   "rccl_ranks": (world if isinstance(comm, sharding.RcclComm) else 0) if world > 1 else None,
                   "slices_per_rank": S},
        "roofline": primary["roofline"],
        "sustained": primary["sustained"],
    }
    # per-rank checksum of one step's outputs on slot 0 (rank r's data depend on r only, so its checksum must be the
    # same in the 1-, 2-, 4- and 8-GPU runs: sharding changes no bit)
    out["rank_checksums"] = [f"{v:016x}" for v in (comm.allgather_u64(primary["checksum"]) if comm else [primary["checksum"]])]
    # slice_checksums[r][j] = sub-batch j ("virtual rank" r * S + j) of rank r: equal across every way of cutting the same total batch
    cols = [(comm.allgather_u64(c_) if comm else [c_]) for c_ in primary["slice_checksums"]]
    out["slice_checksums"] = [[f"{cols[j][r]:016x}" for j in range(S)] for r in range(world)]
    if rank == 0:
        lay_, gains, arr, u, fl = chk
        out["flags"] = {"eigen_path_frac": float(((fl & 4) != 0).mean()), "pinv_branch_frac": float(((fl & 2) != 0).mean()),
                        "truncated_frac": float(((fl & 8) != 0).mean()), "nonfinite_frac": float(((fl & 64) != 0).mean()),
                        "giveups_to_generic_kernel": primary["giveups"]}
        if cb is not None:
            out["cpu_baseline"] = cb
            if ref is None:      # (fallback above) the oracle in this process on the first instances of the GPU's own slot 0
                n = min(NSEC, B)
                ref = np.full((u.shape[0], u.shape[1]), np.nan)
                ref[:n] = oracle_reference(lay_, gains, arr, 0, n)
                ref_idx = range(n)
            out["parity_sample"] = parity_sample(arr, u, ref, ref_idx)
    runs = [(m, args.workload) for m in others]
    if others and args.workload == "physical":
        runs.append((args.dtype, "synthetic"))         # the workload of rounds 1-3 (dense M without structural zeros), for comparison
    if runs:
        out["secondary"] = []
        for other, wl in runs:
            try:
                sec, schk = measure(other, max(24, min(200, args.steps // 4)), max(8, min(24, args.warmup // 4)),
                                    preroll=min(args.preroll, 100), workload=wl, long_leg=False)
            except Exception as e:
                out["secondary"].append({"mode": other, "records_from": wl, "error": str(e)})
                continue
            slay = sec.pop("layout")
            entry = {"mode": other, "records_from": wl, "dtype": sec["arith"], "records": sec["records"], "value": sec["value"],
                     "ms_per_step": sec["ms_per_step"], "kernel": sec["kernel"], "roofline": sec["roofline"]}
            if not args.no_cpu_baseline:                 # the oracle in this process on the first instances of slot 0
                _, sgains, sarr, su, _ = schk
                n = min(NSEC, B)
                full = np.full((su.shape[0], su.shape[1]), np.nan)
                full[:n] = oracle_reference(slay, sgains, sarr, 0, n)
                entry["parity_sample"] = parity_sample(
                    sarr, su, full, range(n), tol=1e-5,
                    note="GPU vs float64 oracle on the same (record-dtype-rounded) records; parity domain per SURVEY.md 8c")
            out["secondary"].append(entry)
    for e_ in out.get("secondary", []):       # rounds 1-3 quoted this workload as the headline: kept at top level under its own name
        if e_.get("records_from") == "synthetic" and e_.get("mode") == args.dtype and "value" in e_:
            out["synthetic_dense_records"] = {"metric": "OSC control steps/sec on synthetic dense records (random SPD M without the tree's zeros)",
                                              "value": e_["value"], "ms_per_step": e_["ms_per_step"], "kernel": e_["kernel"],
                                              "roofline_frac": e_["roofline"]["frac"]}
    if world == 1 and not args.no_from_q:
        out["from_q"] = measure_from_q(BatchedOSC, synth, args, B, local_rank, fq_state, fq_ref)
    if world == 1 and rank == 0 and not args.no_end_to_end:
        lay_, gains_, arr_ = chk[0], chk[1], chk[2]
        out["end_to_end"] = measure_end_to_end(BatchedOSC, lay_, gains_, arr_, MODES[args.dtype][0],
                                               args.kernel if args.kernel >= 0 else MODES[args.dtype][2], local_rank)
    # Flat copies of the nested results: the driver's BENCH record keeps the scalars of `config` / `roofline` / `cpu_baseline` and only the
    # NAMES of every other top-level key.
    cfgd = out["config"]
    if out.get("sustained"):
        cfgd.update(sustained_value=out["sustained"]["value"], sustained_steps=out["sustained"]["steps"],
                    sustained_ms_per_step=out["sustained"]["ms_per_step"])
    if out.get("parity_sample"):
        ps = out["parity_sample"]
        cfgd.update(parity_n_checked=ps["n"], parity_max_rel_err=ps["max_rel_err"], parity_n_over_tol=ps["n_over_tol"],
                    parity_tolerance=ps["tolerance"], parity_n_outside_domain=ps.get("n_outside_parity_domain"),
                    parity_n_examined_for_the_domain=ps.get("n_examined_for_the_domain"))
    if out.get("flags"):
        cfgd.update(eigen_path_frac=out["flags"]["eigen_path_frac"], truncated_frac=out["flags"]["truncated_frac"],
                    giveups_to_generic_kernel=out["flags"]["giveups_to_generic_kernel"])
    fq = out.get("from_q") or {}
    if "value" in fq:
        cfgd.update(from_q_value=fq["value"], from_q_ms_per_step=fq.get("ms_per_step"), from_q_kernel=fq.get("kernel"),
                    from_q_roofline_frac_fp64_valu=(fq.get("roofline") or {}).get("frac"),
                    from_q_parity_max_rel_err=(fq.get("parity_sample") or {}).get("max_rel_err"))
    for e_ in out.get("secondary", []):
        if "value" in e_ and e_.get("records_from") == args.workload:
            cfgd[f"secondary_{e_['mode']}_value"] = e_["value"]
            cfgd[f"secondary_{e_['mode']}_roofline_frac"] = e_["roofline"]["frac"]
            if e_.get("parity_sample"):
                cfgd[f"secondary_{e_['mode']}_parity_n_over_tol"] = e_["parity_sample"]["n_over_tol"]
    if out.get("synthetic_dense_records"):
        cfgd.update(synthetic_dense_value=out["synthetic_dense_records"]["value"],
                    synthetic_dense_roofline_frac=out["synthetic_dense_records"]["roofline_frac"])
    e2e = out.get("end_to_end") or {}
    if isinstance(e2e.get("generate_batched"), dict) and "value" in e2e["generate_batched"]:
        cfgd.update(end_to_end_host_arrays_value=e2e["generate_batched"]["value"], end_to_end_pcie_GBps=e2e["generate_batched"].get("pcie_GBps"))
    if isinstance(e2e.get("tick_b1_us"), dict):
        cfgd["end_to_end_tick_b1_us"] = e2e["tick_b1_us"].get("median")
    for key, flat in (("generate_batched_float32", "end_to_end_host_arrays_f32_value"), ("upload_raw_sparse_step", "end_to_end_raw_sparse_qM_value"),
                      ("upload_raw_sparse_step_float32", "end_to_end_raw_sparse_qM_f32_value")):
        if isinstance(e2e.get(key), dict) and "value" in e2e[key]:
            cfgd[flat] = e2e[key]["value"]
            cfgd[flat.replace("_value", "_pcie_GBps")] = e2e[key].get("pcie_GBps")
    # The driver's record keeps the first 24 keys of `config` and of `roofline`: the evidence scalars go first, the prose last
    # (tests/test_bench_host.py::test_evidence_scalars_lead_the_line).
    out["config"] = ordered_first(cfgd, CONFIG_FIRST)
    out["roofline"] = ordered_first(out["roofline"], ROOFLINE_FIRST)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm:
        comm.close()


if __name__ == "__main__":
    main()
