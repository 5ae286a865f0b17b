"""N > 1 path on CPU: two ranks shard one batch, each computes its slice, and the gathered result
equals the unsharded one bit for bit; the throughput reduction takes sum(steps) / max(time); the
RCCL unique id travels from rank 0 to the others through the rendezvous file.  The collective itself
is RCCL on the GPU box (irlosc_bench_allreduce, tests/test_gpu_parity.py); here a gloo communicator
with the same reduce() contract stands in, so that sharding.reduce_throughput and bench.py's N > 1
bookkeeping run with world_size 2.  (The per-slice compute here is the oracle: the HIP path under a
shard is checked bit for bit on the GPU box, test_row16_ragged_batches_and_sharding_bit_exact.)"""
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


class GlooComm:
    """Same contract as sharding.RcclComm.reduce / allgather_u64, over torch.distributed gloo (test stand-in)."""

    def reduce(self, steps, elapsed):
        t = torch.tensor([float(steps)], dtype=torch.float64)
        e = torch.tensor([float(elapsed)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        return float(t[0]), float(e[0])

    def allgather_u64(self, mine):
        out = [torch.zeros(1, dtype=torch.int64) for _ in range(dist.get_world_size())]
        dist.all_gather(out, torch.tensor([mine - (1 << 64) if mine >= (1 << 63) else mine], dtype=torch.int64))
        return [int(v[0]) % (1 << 64) for v in out]


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
    comm = GlooComm()
    steps, elapsed, rate = sharding.reduce_throughput(hi - lo, 1.0 + rank, comm)   # rank 1 is "slower"
    sums = comm.allgather_u64(sharding.checksum_u64(u))
    # the unique-id hand-over bench.py uses before irlosc_comm_create: rank 0 publishes 128 bytes, the others wait
    uid = sharding.exchange_bytes(rank, bytes(range(128)) if rank == 0 else None, 128,
                                  sharding.rendezvous_path(f"test_{port}"))
    assert uid == bytes(range(128))
    if rank == 0:
        np.save(os.path.join(out_dir, "sums.npy"), np.array(sums, dtype=np.uint64))
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
    sums = np.load(tmp_path / "sums.npy")
    assert [int(v) for v in sums] == [sharding.checksum_u64(ref[lo:hi]) for lo, hi in
                                      (sharding.shard_range(B, world, r) for r in range(world))]


def test_reduce_throughput_single_process_and_env_defaults():
    assert sharding.reduce_throughput(10, 2.0) == (10.0, 2.0, 5.0)
    a = np.arange(12, dtype=np.float64)
    assert sharding.checksum_u64(a) == sharding.checksum_u64(a.copy()) != sharding.checksum_u64(a[::-1])


def _filecomm_worker(rank, world, tag, q):
    from irl_control_amd import sharding
    c = sharding.FileComm(rank, world, tag=tag, timeout_s=60.0)
    out = []
    for it in range(5):                                   # several operations in a row: files of old operations are recycled
        out.append(c.reduce(10.0 * (rank + 1) + it, 0.5 + rank + it))
    c.barrier()
    out.append(c.allgather_u64((0xdeadbeef00000000 + rank) & (2 ** 64 - 1)))
    c.close()
    q.put((rank, out))


def test_filecomm_fallback_three_ranks(tmp_path, monkeypatch):
    """bench.py's fallback when RCCL cannot be brought up: sum / max / all-gather through files, three processes."""
    import multiprocessing as mp
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    ps = [ctx.Process(target=_filecomm_worker, args=(r, world, "t1", q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
    for r in range(world):
        for it in range(5):
            assert res[r][it] == (10.0 * 6 + 3 * it, 0.5 + 2 + it)
        assert res[r][5] == [0xdeadbeef00000000 + k for k in range(world)]
    import os
    import stat
    from irl_control_amd import sharding
    d = sharding.rendezvous_dir("t1")                      # TMPDIR is tmp_path here: the launch's private directory
    assert os.path.dirname(d) == str(tmp_path) and stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    left = [f for f in os.listdir(d) if "_file_op" in f]
    assert 0 < len(left) <= world                          # only the last operation's files stay


def test_rendezvous_dir_refuses_a_planted_path(tmp_path, monkeypatch):
    """A shared /tmp: somebody else's directory, a world-writable one, or a symlink under the expected name is refused."""
    import os
    from irl_control_amd import sharding
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    planted = tmp_path / f"irlosc_{os.getuid()}_bad"
    planted.mkdir(mode=0o777)
    os.chmod(planted, 0o777)
    with pytest.raises(PermissionError):
        sharding.rendezvous_dir("bad")
    target = tmp_path / "elsewhere"
    target.mkdir()
    os.symlink(target, tmp_path / f"irlosc_{os.getuid()}_link")
    with pytest.raises(PermissionError):
        sharding.rendezvous_dir("link")
    assert os.path.isdir(sharding.rendezvous_dir("fine"))



def _make_comm_worker(rank, world, tag, q):
    from irl_control_amd import sharding
    comm, note = sharding.make_comm(rank, world, 0, tag=tag, init_timeout_s=60.0)
    total, worst = comm.reduce(rank + 1.0, 0.25 * (rank + 1))
    comm.close()
    q.put((rank, type(comm).__name__, note, total, worst))


def test_make_comm_agrees_on_the_file_transport_when_rccl_is_unavailable(tmp_path, monkeypatch):
    """No GPU here, so RCCL cannot come up on any rank: every rank must land on FileComm (the choice is collective: a
    job with mixed transports would hang) and the reduction must work through it."""
    import multiprocessing as mp
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_make_comm_worker, args=(r, 2, "mc", q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert [r[1] for r in res] == ["FileComm", "FileComm"]
    assert all("files in TMPDIR" in r[2] and "rank(s) [0, 1]" in r[2] for r in res)
    assert all(r[3] == 3.0 and r[4] == 0.5 for r in res)
