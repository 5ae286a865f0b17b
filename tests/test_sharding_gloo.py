"""N > 1 path on CPU: two gloo ranks shard one batch, each computes its slice, and the gathered
result equals the unsharded one bit for bit; the throughput reduction takes sum(steps) / max(time).
(The per-slice compute here is the oracle — this test is about the sharding logic, which is what
bench.py --gpus N uses around the HIP path.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from irl_control_amd import sharding, synth
from oracle import osc_oracle


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 16, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lay, gains, g = synth.make_batch("k13", B, seed=11)
    lo, hi = sharding.shard_range(B, world, rank)
    u = osc_oracle.generate_batch(lay.as_oracle_dict(), gains, g["M"], g["J"], g["dq"], g["bias"],
                                  g["ee_pose"], g["tgt_pose"], idx=range(lo, hi))[lo:hi]
    sizes = [sharding.shard_range(B, world, r) for r in range(world)]
    parts = [torch.zeros((h - l, 25), dtype=torch.float64) for l, h in sizes]
    dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(u))) if len(set(h - l for l, h in sizes)) == 1 else None
    steps, elapsed, rate = sharding.reduce_throughput(hi - lo, 1.0 + rank)   # rank 1 is "slower"
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), torch.cat(parts).numpy())
        np.save(os.path.join(out_dir, "rate.npy"), np.array([steps, elapsed, rate]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_is_bit_identical(tmp_path):
    B, world = 24, 2
    mp.spawn(_worker, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    lay, gains, g = synth.make_batch("k13", B, seed=11)
    ref = osc_oracle.generate_batch(lay.as_oracle_dict(), gains, g["M"], g["J"], g["dq"], g["bias"],
                                    g["ee_pose"], g["tgt_pose"])
    got = np.load(tmp_path / "gathered.npy")
    assert np.array_equal(got, ref)
    steps, elapsed, rate = np.load(tmp_path / "rate.npy")
    assert steps == B and elapsed == 2.0 and rate == B / 2.0
