#!/usr/bin/env python3
"""bench.py — OSC control steps/sec on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|f64] [--batch B]

A "step" is one pass of the OSC hot path over one resident batch of B = 65 536 synthetic Dual-UR5
instances per GPU (n = 25 joints, k = 13 task rows: both arms xyz+abg and the base yaw, gravity and
null-space terms on; BASELINE.json configs[2]).  Inputs are resident in HBM before the timed
region; `n_slots` distinct batches are rotated so that successive launches do not re-hit the
256 MiB Infinity Cache.  Instances shard across GPUs with no data-path collective (weak scaling:
B per GPU is fixed); torch.distributed (RCCL) is used only for the barrier and the max-over-ranks
of the elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


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


def cpu_baseline(lay, gains, arrays, seconds_target=12.0):
    """Time the oracle (NumPy restatement of the reference, 1 thread) on a bounded sample."""
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
        nsamp = int(max(512, min(a64["M"].shape[0] - 576, seconds_target / (dt / 512))))
        t0 = time.perf_counter(); ref = run(range(576, 576 + nsamp)); dt = time.perf_counter() - t0
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    out = dict(value=nsamp / dt, unit="steps/s", cores=1, kind="port",
               sample=f"{nsamp} of the {a64['M'].shape[0]} instances of slot 0, oracle/osc_oracle.py "
                      f"(float64 NumPy, OPENBLAS threads=1), {dt:.1f} s")
    # a stronger CPU figure next to it: the same arithmetic with the batch axis inside NumPy's stacked LAPACK
    # calls (no per-instance interpreter overhead), BLAS threads left at the library default
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
        out["vectorised"] = dict(value=nb / dtb, unit="steps/s", cores=os.cpu_count(),
                                 sample=f"{nb} instances, oracle/osc_oracle_batched.py (stacked np.linalg calls), {dtb:.1f} s")
    except Exception as e:                                  # the baseline must never break the bench line
        out["vectorised"] = dict(error=str(e))
    return out, ref, range(576, 576 + nsamp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--preroll", type=int, default=2000,
                    help="extra UNTIMED steps before the warm-up: after the idle set-up phase the power management needs ~50 ms of load to settle")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--layout", default="k13")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic, 2 group")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary-dtype measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from irl_control_amd import BatchedOSC, sharding, synth

    def measure(dtype_name, steps, warmup, with_check, preroll):
        dt = np.float32 if dtype_name == "f32" else np.float64
        esz = 4 if dtype_name == "f32" else 8
        B = args.batch
        lay = synth.make_layout(args.layout)
        osc = BatchedOSC(lay, B, dtype=dt, hip_device=local_rank, n_slots=args.slots, kernel=args.kernel)
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
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ms_total, ms_kernel = osc.step_resident(steps)       # HIP events on the library's own stream
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        total_steps, elapsed, rate = sharding.reduce_throughput(B * steps, elapsed, device="cuda")
        if world > 1:
            mk = torch.tensor([ms_kernel], device="cuda", dtype=torch.float64)
            dist.all_reduce(mk, op=dist.ReduceOp.MAX)
            ms_kernel = float(mk[0])
        bytes_step = algorithmic_bytes(lay.n, lay.k, lay.ndev, lay.admittance, esz) * B
        spl = osc.steps_per_launch              # the throughput path chains this many steps into one launch
        bytes_launch = bytes_step * spl
        # dominant kernel alone (the fused train launch of the group path / the generic kernel), HIP events on its stream
        ms_dom = osc.time_dominant_kernel(min(200, max(10 * spl, steps // 2)))
        if world > 1:
            md = torch.tensor([ms_dom], device="cuda", dtype=torch.float64)
            dist.all_reduce(md, op=dist.ReduceOp.MAX)
            ms_dom = float(md[0])
        achieved = bytes_launch / (ms_dom * 1e-3) / 1e9
        res = dict(value=rate, ms_per_step=elapsed / steps * 1e3, kernel=osc.kernel_name,
                   roofline=dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                                 frac=achieved / HBM_PEAK_GBS, traffic=measured_traffic(osc.kernel_name),
                                 kernel=osc.kernel_name + ((":fused(stage 1 of %d chained steps + riding stage 2 of the previous launch)" % spl) if "group" in osc.kernel_name else ""),
                                 kernel_ms=ms_dom, steps_per_launch=spl, step_ms_events=ms_kernel,
                                 whole_step_achieved=bytes_step / (ms_kernel * 1e-3) / 1e9,
                                 algorithmic_bytes_per_launch=bytes_launch))
        check = None
        if with_check:
            osc.step(slot=0)
            u, fl = osc.download(B)
            check = (lay, gains, slot0, u, fl)
        osc.close()
        return res, check

    primary, chk = measure(args.dtype, args.steps, args.warmup, with_check=(rank == 0), preroll=args.preroll)
    out = {
        "metric": "OSC control steps/sec", "value": primary["value"], "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": primary["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.batch} Dual-UR5 instances per GPU (BASELINE configs[2]), n=25 joints, "
                               f"layout {args.layout}: k=13 task rows (both arms xyz+abg, base yaw), gravity + "
                               f"null-space on, inputs resident in HBM, {args.slots} rotating batches",
                   "instances_per_gpu": args.batch, "layout": args.layout, "kernel": primary["kernel"],
                   "preroll_steps": args.preroll, "steps_per_launch": primary["roofline"]["steps_per_launch"],
                   "sharding": f"{world} x independent shards, no data-path collective"},
        "roofline": primary["roofline"],
    }
    if rank == 0 and chk is not None:
        lay, gains, arr, u, fl = chk
        out["flags"] = {"eigen_path_frac": float(((fl & 4) != 0).mean()), "pinv_branch_frac": float(((fl & 2) != 0).mean()),
                        "truncated_frac": float(((fl & 8) != 0).mean()), "nonfinite_frac": float(((fl & 64) != 0).mean())}
        if world == 1 and not args.no_cpu_baseline:
            cb, ref, idx = cpu_baseline(lay, gains, arr)
            out["cpu_baseline"] = cb
            idx = np.asarray(list(idx))
            err = np.max(np.abs(u[idx].astype(np.float64) - ref[idx]), axis=1) / np.max(np.abs(ref[idx]), axis=1)
            out["parity_sample"] = {"n": int(len(idx)), "median_rel_err": float(np.median(err)),
                                    "p99_rel_err": float(np.quantile(err, 0.99)),
                                    "note": "GPU vs float64 oracle on the same (dtype-rounded) inputs"}
    if world == 1 and not args.no_secondary:
        other = "f64" if args.dtype == "f32" else "f32"
        sec, _ = measure(other, max(20, min(200, args.steps // 4)), max(5, min(20, args.warmup // 4)), with_check=False,
                         preroll=min(args.preroll, 40))
        out["secondary"] = {"dtype": other, "value": sec["value"], "ms_per_step": sec["ms_per_step"],
                            "kernel": sec["kernel"], "roofline": sec["roofline"]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
