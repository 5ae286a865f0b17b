#!/usr/bin/env python3
"""bench.py — OSC control steps/sec on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f64|mixed|f32] [--batch B] [--layout k13|k7|k12_admit]

A "step" is one pass of the OSC hot path over one resident batch of B synthetic Dual-UR5 instances per GPU
(default: B = 65 536, n = 25 joints, layout k13 = both arms xyz+abg and the base yaw, gravity and null-space
terms on: BASELINE.json configs[2]).  Inputs are resident in HBM before the timed region; `--slots` distinct
batches are rotated so that successive launches do not re-hit the 256 MiB Infinity Cache.

--dtype names the ARITHMETIC of the measured path:
  f64    float64 records, float64 arithmetic (osc_row16 kernel)  - the reference's precision, meets north_star's 1e-5
  mixed  float32 records, float64 arithmetic (osc_row16 kernel)  - BASELINE configs[2]'s fp32 storage at the 1e-5 bar
  f32    float32 records, float32 arithmetic (osc_group kernel)  - fastest; error ~ eps32 * cond(J M^-1 J^T)
The JSON line's "dtype" is the arithmetic type ("f64" for f64 and mixed); config.records names the storage.

Instances shard across GPUs with no data-path collective (weak scaling: B per GPU is fixed; `--total-batch T`
divides T over the ranks instead, e.g. BASELINE configs[3]: --gpus 8 --total-batch 262144).  One process per GPU,
launched as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`; the launcher only provides
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.  No PyTorch here: the barrier, the sum of steps, the max of the elapsed
time and the all-gather of per-rank output checksums are RCCL calls behind the C ABI (irlosc_bench_allreduce,
irlosc_comm_allgather_u64), and the device synchronisation is hipDeviceSynchronize (irlosc_device_sync).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MODES = {   # --dtype -> (record dtype, arithmetic label, kernel id)
    "f64": (np.float64, "f64", 0),
    "mixed": (np.float32, "f64", 3),
    "f32": (np.float32, "f32", 0),
}


def algorithmic_bytes(n, k, ndev, admittance, esz):
    """s*(n^2 + k*n + 2n + 14*ndev + [6*ndev] + n_out), n_out = n."""
    return esz * (n * n + k * n + 2 * n + 14 * ndev + (6 * ndev if admittance else 0) + n)


def measured_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/hbm_traffic.json: FETCH_SIZE x2-corrected + WRITE_SIZE), or None."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel_name, {}).get("traffic_bytes_per_launch")
    except (OSError, ValueError):
        return None


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


def _cpu_worker(args):
    """One worker = one core: the op-for-op oracle over its share of instance ids, BLAS pinned to one thread."""
    wid, nworkers, seconds = args
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except ImportError:
        ctx = None
    from oracle import osc_oracle
    a, od, gains = _CPU["a"], _CPU["od"], _CPU["gains"]
    Btot = a["M"].shape[0]
    done, t0 = 0, time.perf_counter()
    chunk = 64
    pos = (wid * (Btot // nworkers)) % Btot
    while time.perf_counter() - t0 < seconds:
        idx = [(pos + i) % Btot for i in range(chunk)]
        osc_oracle.generate_batch(od, gains, a["M"], a["J"], a["dq"], a["bias"], a["ee_pose"], a["tgt_pose"],
                                  a.get("wrench"), a.get("tgt_vel"), idx=idx)
        done += chunk
        pos = (pos + chunk) % Btot
    dt = time.perf_counter() - t0
    if ctx is not None:
        ctx.__exit__(None, None, None)
    return done, dt


def cpu_baseline(lay, gains, arrays, seconds_single=8.0, seconds_all=8.0):
    """-> (cpu_baseline dict, reference outputs, their instance ids).  Three figures:
    all host cores (one oracle worker per core, os.cpu_count() stated) = the headline; one core; and the stacked-LAPACK
    restatement as a stronger single-process comparator."""
    from oracle import osc_oracle
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        threadpool_limits = None
    a64 = {k: v.astype(np.float64) for k, v in arrays.items()}
    od = lay.as_oracle_dict()

    def run(idx):
        return osc_oracle.generate_batch(od, gains, a64["M"], a64["J"], a64["dq"], a64["bias"], a64["ee_pose"],
                                         a64["tgt_pose"], a64.get("wrench"), a64.get("tgt_vel"), idx=idx)
    ctx = threadpool_limits(limits=1) if threadpool_limits else None
    try:
        run(range(0, 64))                                   # warm-up
        t0 = time.perf_counter(); run(range(64, 576)); dt = time.perf_counter() - t0
        nsamp = int(max(512, min(a64["M"].shape[0] - 576, seconds_single / (dt / 512))))
        t0 = time.perf_counter(); ref = run(range(576, 576 + nsamp)); dt = time.perf_counter() - t0
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    single = dict(value=nsamp / dt, unit="steps/s", cores=1,
                  sample=f"{nsamp} of the {a64['M'].shape[0]} instances of slot 0, oracle/osc_oracle.py "
                         f"(float64 NumPy, OPENBLAS threads=1), {dt:.1f} s")
    out = dict(single)
    out["kind"] = "port"
    # whole host: one worker per core, fork (the arrays are inherited, not pickled)
    ncores = os.cpu_count() or 1
    try:
        _CPU.update(a=a64, od=od, gains=gains)
        with mp.get_context("fork").Pool(ncores) as pool:
            t0 = time.perf_counter()
            res = pool.map(_cpu_worker, [(w, ncores, seconds_all) for w in range(ncores)])
            wall = time.perf_counter() - t0
        total = sum(r[0] for r in res)
        out = dict(value=total / wall, unit="steps/s", cores=ncores, kind="port",
                   sample=f"{total} oracle steps by {ncores} worker processes (os.cpu_count() = {ncores}, one per core, "
                          f"BLAS threads = 1 each) in {wall:.1f} s wall incl. pool start-up, instances of slot 0",
                   single_core=single)
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
        out["vectorised"] = dict(value=nb / dtb, unit="steps/s", cores="one process, BLAS threads at the library default",
                                 sample=f"{nb} instances, oracle/osc_oracle_batched.py (stacked np.linalg calls), {dtb:.1f} s")
    except Exception as e:
        out["vectorised"] = dict(error=str(e))
    return out, ref, range(576, 576 + nsamp)


def parity_sample(lay, arr, u, ref, idx):
    """GPU vs float64 oracle on the same (record-dtype-rounded) inputs; the instances over 1e-5 are then classified
    against the parity domain (the reference's own answer well defined, no singular value within
    1 % of the pinv cut)."""
    from oracle import osc_oracle
    idx = np.asarray(list(idx))
    err = np.max(np.abs(u[idx].astype(np.float64) - ref[idx]), axis=1) / np.max(np.abs(ref[idx]), axis=1)
    over = idx[err > 1e-5]
    in_dom = 0
    for b in over[:2000]:
        Mx, Minv, Mxi, det = osc_oracle.task_inertia(arr["J"][b].astype(np.float64), arr["M"][b].astype(np.float64))
        s = np.linalg.svd(Mxi, compute_uv=False)
        if abs(det) >= 1e-4:
            ok = s[-1] > 1e-12 * s[0]
        else:
            ok = not np.any(np.abs(s / s[0] / 1e-5 - 1.0) < 1e-2)
        in_dom += bool(ok)
    return {"n": int(len(idx)), "median_rel_err": float(np.median(err)), "p99_rel_err": float(np.quantile(err, 0.99)),
            "max_rel_err": float(err.max()), "n_over_1e-5": int(len(over)), "n_over_1e-5_in_parity_domain": int(in_dom),
            "note": "GPU vs float64 oracle on the same (record-dtype-rounded) inputs; parity domain per SURVEY.md 8c"}


def measure_from_q(BatchedOSC, synth, args, B, local_rank):
    """The path from joint coordinates: per step the rigid-body front end computes M, J,
    bias and the EE poses from resident (qpos, qvel) on the GPU, then the OSC step runs on them.  What a host-side
    simulator would have to ship per step otherwise is the 8.5 KB of records (PCIe ceiling ~63 GB/s / 8 536 B = 7.4e6
    steps/s in float64, 1.5e7 in float32)."""
    from irl_control_amd.rigid_body import RigidBodyModel
    try:
        dt, arith, kern = MODES[args.dtype]
        lay = synth.make_layout(args.layout)
        model = RigidBodyModel.load("dual_ur5")
        osc = BatchedOSC(lay, B, dtype=dt, hip_device=local_rank, n_slots=args.slots, kernel=kern)
        osc.set_model(model)
        rng = np.random.default_rng(20241008 + 77)
        _, gains, arr = synth.make_batch(args.layout, B, seed=20241008 + 2000, dtype=dt)
        osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
        for s in range(args.slots):
            qpos, qvel = model.random_state(rng, B)
            osc.upload_q(qpos, qvel, slot=s)
            osc.frontend(slot=s)
            osc.set_targets(arr["tgt_pose"], arr.get("tgt_vel"), slot=s)
        steps = max(20, min(200, args.steps // 4))
        osc.step_resident_from_q(20)
        osc.device_sync()
        t0 = time.perf_counter()
        ms_total, ms_step = osc.step_resident_from_q(steps)
        osc.device_sync()
        el = time.perf_counter() - t0
        _, ms_osc = osc.step_resident(steps)
        esz = np.dtype(dt).itemsize
        res = dict(value=B * steps / el, unit="steps/s", ms_per_step=el / steps * 1e3, ms_per_step_events=ms_step,
                   ms_osc_step_alone=ms_osc, ms_front_end=ms_step - ms_osc, kernel=osc.frontend_name + " + " + osc.kernel_name,
                   input_bytes_per_step_per_instance=2 * lay.n * 8 + 7 * lay.ndev * esz,
                   note="per step: rigid-body front end (FK, EE Jacobians, CRBA, RNEA) from resident (qpos, qvel), then the OSC "
                        "step on the records it wrote; nothing crosses PCIe")
        osc.close()
        return res
    except Exception as e:                              # never break the bench line
        return dict(error=str(e))


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
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--layout", default="k13")
    ap.add_argument("--kernel", type=int, default=-1, help="override: 0 auto, 1 generic, 2 group, 3 row16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other arithmetic / storage variants")
    ap.add_argument("--no-from-q", action="store_true", help="skip the joint-coordinates path (front end + step)")
    args = ap.parse_args()

    from irl_control_amd import BatchedOSC, sharding, synth
    rank, world, local_rank = sharding.env_world()
    if "IRLOSC_BENCH_DEVICE" in os.environ:                  # test hook: several ranks on ONE GPU (then RCCL refuses the duplicate
        local_rank = int(os.environ["IRLOSC_BENCH_DEVICE"])  # device and the file-based reduction is what gets exercised)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 "
                     "--master-port P bench.py --gpus N ...")
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    B = args.total_batch // world if args.total_batch else args.batch
    comm, comm_note = None, "RCCL (through the C ABI)"
    if world > 1:
        try:
            comm = sharding.RcclComm(rank, world, local_rank)
        except Exception as e:                                # still report the line; say how it was reduced
            comm = sharding.FileComm(rank, world)
            comm_note = f"files in TMPDIR (RCCL could not be brought up: {e})"

    def measure(mode, steps, warmup, with_check, preroll):
        dt, arith, kern = MODES[mode]
        if args.kernel >= 0:
            kern = args.kernel
        esz = np.dtype(dt).itemsize
        lay = synth.make_layout(args.layout)
        osc = BatchedOSC(lay, B, dtype=dt, hip_device=local_rank, n_slots=args.slots, kernel=kern)
        slot0 = None
        for s in range(args.slots):
            _, gains, arr = synth.make_batch(args.layout, B, seed=20241008 + 1000 * 2 + 17 * s + 101 * rank, dtype=dt)
            osc.upload(arr["M"], arr["J"], arr["dq"], arr["bias"], arr["ee_pose"], arr.get("wrench"), slot=s)
            osc.set_targets(arr["tgt_pose"], arr.get("tgt_vel"), slot=s)
            if s == 0:
                slot0 = arr
                osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"],
                              gains["null_kv"])
            else:
                del arr
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
        bytes_step = algorithmic_bytes(lay.n, lay.k, lay.ndev, lay.admittance, esz) * B
        spl = osc.steps_per_launch              # the fp32 group path chains this many steps into one launch
        bytes_launch = bytes_step * spl
        # dominant kernel alone, HIP events on its stream (fp32 group: the fused train launch; else one step's kernel)
        ms_dom = osc.time_dominant_kernel(min(200, max(10 * spl, steps // 2)))
        if comm:
            ms_dom = comm.reduce(0.0, ms_dom)[1]
            ms_kernel = comm.reduce(0.0, ms_kernel)[1]
        achieved = bytes_launch / (ms_dom * 1e-3) / 1e9
        kname = osc.kernel_name
        note = ""
        if "group" in kname:
            note = ":fused(stage 1 of %d chained steps + riding stage 2 of the previous launch)" % spl
        elif "row16" in kname:
            note = ":step(all instances incl. the in-kernel truncated-pinv stage; + the give-up list launch)"
        res = dict(value=rate, ms_per_step=elapsed / steps * 1e3, kernel=kname, mode=mode, arith=arith,
                   records="float64" if esz == 8 else "float32", layout=lay,
                   roofline=dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                                 frac=achieved / HBM_PEAK_GBS, traffic=measured_traffic(kname),
                                 kernel=kname + note, kernel_ms=ms_dom, steps_per_launch=spl, step_ms_events=ms_kernel,
                                 whole_step_achieved=bytes_step / (ms_kernel * 1e-3) / 1e9,
                                 algorithmic_bytes_per_launch=bytes_launch,
                                 algorithmic_bytes_per_step_per_instance=bytes_step // B))
        check = None
        osc.step(slot=0)
        u, fl = osc.download(B)
        res["checksum"] = sharding.checksum_u64(u)
        if with_check:
            check = (lay, gains, slot0, u, fl)
        osc.close()
        return res, check

    # CPU baseline first: its worker processes are forked before this process has initialised the HIP runtime
    cb = None
    if world == 1 and not args.no_cpu_baseline:
        lay0, gains0, arr0 = synth.make_batch(args.layout, B, seed=20241008 + 1000 * 2 + 101 * rank, dtype=MODES[args.dtype][0])
        cb, ref, ref_idx = cpu_baseline(lay0, gains0, arr0)
        del arr0
    primary, chk = measure(args.dtype, args.steps, args.warmup, with_check=(rank == 0), preroll=args.preroll)
    lay = primary.pop("layout")
    names = ", ".join(f"{nm}:{r}" for nm, r in zip(lay.dev_names, lay.dev_rows))
    out = {
        "metric": "OSC control steps/sec", "value": primary["value"], "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": primary["ms_per_step"],
        "higher_is_better": True, "scaling": "strong" if args.total_batch else "weak", "vs_baseline": None,
        "dtype": primary["arith"], "data": "synthetic",
        "config": {"workload": f"{B} Dual-UR5 instances per GPU ({baseline_config_of(args.layout, B, world, args.dtype)}), "
                               f"n={lay.n} joints, layout {args.layout}: k={lay.k} task rows over {lay.ndev} target devices "
                               f"({names}), gravity {'on' if lay.use_g else 'off'}, null-space {'on' if lay.nullspace else 'off'}, "
                               f"admittance wrench term {'on' if lay.admittance else 'off'}, {primary['records']} records, "
                               f"{primary['arith']} arithmetic, inputs resident in HBM, {args.slots} rotating batches",
                   "instances_per_gpu": B, "layout": args.layout, "k": lay.k, "ndev": lay.ndev,
                   "admittance": bool(lay.admittance), "records": primary["records"], "arithmetic": primary["arith"],
                   "kernel": primary["kernel"], "preroll_steps": args.preroll,
                   "steps_per_launch": primary["roofline"]["steps_per_launch"],
                   "sharding": f"{world} x independent shards, no data-path collective; barrier and final sum(steps) / "
                               f"max(elapsed) reduction: {comm_note if world > 1 else 'single process'}"},
        "roofline": primary["roofline"],
    }
    # per-rank checksum of one step's outputs on slot 0 (rank r's data depend on r only, so its checksum must be the
    # same in the 1-, 2-, 4- and 8-GPU runs: sharding changes no bit)
    out["rank_checksums"] = [f"{v:016x}" for v in (comm.allgather_u64(primary["checksum"]) if comm else [primary["checksum"]])]
    if rank == 0 and chk is not None:
        lay_, gains, arr, u, fl = chk
        out["flags"] = {"eigen_path_frac": float(((fl & 4) != 0).mean()), "pinv_branch_frac": float(((fl & 2) != 0).mean()),
                        "truncated_frac": float(((fl & 8) != 0).mean()), "nonfinite_frac": float(((fl & 64) != 0).mean())}
        if cb is not None:
            out["cpu_baseline"] = cb
            out["parity_sample"] = parity_sample(lay_, arr, u, ref, ref_idx)
    if world == 1 and not args.no_secondary:
        out["secondary"] = []
        for other in [m for m in ("f64", "mixed", "f32") if m != args.dtype]:
            try:
                sec, _ = measure(other, max(20, min(200, args.steps // 4)), max(5, min(20, args.warmup // 4)),
                                 with_check=False, preroll=min(args.preroll, 100))
            except Exception as e:                       # e.g. a layout without a group kernel
                out["secondary"].append({"mode": other, "error": str(e)})
                continue
            sec.pop("layout")
            out["secondary"].append({"mode": other, "dtype": sec["arith"], "records": sec["records"], "value": sec["value"],
                                     "ms_per_step": sec["ms_per_step"], "kernel": sec["kernel"], "roofline": sec["roofline"]})
    if world == 1 and not args.no_from_q:
        out["from_q"] = measure_from_q(BatchedOSC, synth, args, B, local_rank)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm:
        comm.close()


if __name__ == "__main__":
    main()
