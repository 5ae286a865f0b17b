"""Multi-GPU sharding of a batch of independent robot instances.

The OSC path has no cross-instance term (/root/reference/irl_control/osc.py:120-210 touches one
robot), so a node runs one process per GPU, each owning a contiguous slice of the batch; there is
NO data-path collective.  RCCL (over xGMI), driven through the C ABI (`irlosc_comm_*`,
`irlosc_bench_allreduce`: include/irlosc.h), is used only for the "final throughput reduction" of
BASELINE.json: sum of the steps done, max of the elapsed time, and an all-gather of per-shard output
checksums.  No PyTorch: the launcher (`python -m torch.distributed.run`) only provides RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment.
"""
import ctypes as C
import os
import stat
import threading
import time
import zlib
from typing import List, Optional, Tuple

import numpy as np

from . import _lib


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` instances for `rank`; sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def env_world() -> Tuple[int, int, int]:
    """(rank, world, local_rank) as the launcher exports them; (0, 1, 0) when run directly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def checksum_u64(a: np.ndarray) -> int:
    """Order-sensitive 64-bit checksum of an array's bytes (two CRC32 halves): equal iff bit-identical in practice."""
    b = np.ascontiguousarray(a).view(np.uint8)
    h = len(b) // 2
    return (zlib.crc32(b[:h].tobytes()) << 32) | zlib.crc32(b[h:].tobytes())


def rendezvous_dir(tag: Optional[str] = None) -> str:
    """Private directory of THIS launch for the rendezvous files (RCCL unique id, transport markers, FileComm): every
    worker of one launcher shares its parent process and MASTER_PORT (a self-launching bench.py hands its workers a
    random IRLOSC_RDV_TAG instead), so both go into the name.  Created 0700; an existing path is accepted only if it is
    a real directory owned by this user with no group / other access (a shared /tmp must not let another user plant or
    redirect the files)."""
    tag = tag or os.environ.get("IRLOSC_RDV_TAG") or f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
    d = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"irlosc_{os.getuid()}_{tag}")
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"rendezvous directory {d} is not a private directory of uid {os.getuid()}")
    return d


def rendezvous_path(tag: Optional[str] = None) -> str:
    """Where rank 0 leaves the RCCL unique id for the other ranks of this launch."""
    return os.path.join(rendezvous_dir(tag), "rccl_id")


def exchange_bytes(rank: int, payload: Optional[bytes], nbytes: int, path: str, timeout_s: float = 120.0) -> bytes:
    """Rank 0 publishes `payload` (atomic rename), every other rank waits for it.  One node only (shared /tmp)."""
    if rank == 0:
        assert payload is not None and len(payload) == nbytes
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, path)
        return payload
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                data = f.read()
            if len(data) == nbytes:
                return data
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: no rendezvous file {path} after {timeout_s:.0f} s")
        time.sleep(0.01)


class RcclComm:
    """RCCL communicator of one process-per-GPU job, through libirlosc (no torch)."""

    def __init__(self, rank: int, world: int, hip_device: int, tag: Optional[str] = None, rdv_timeout_s: float = 120.0):
        self.lib = _lib.load()
        self.rank, self.world = rank, world
        path = rendezvous_path(tag)
        uid = None
        if rank == 0:
            buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
            rc = self.lib.irlosc_comm_unique_id(buf)
            if rc != 0:
                raise _lib.IrloscError(f"irlosc_comm_unique_id failed ({rc}): {self.lib.irlosc_comm_last_error(None).decode()}")
            uid = bytes(buf)
        uid = exchange_bytes(rank, uid, _lib.COMM_ID_BYTES, path, timeout_s=rdv_timeout_s)
        h = C.c_void_p()
        idbuf = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(uid)
        rc = self.lib.irlosc_comm_create(hip_device, rank, world, idbuf, C.byref(h))
        if rc != 0:
            raise _lib.IrloscError(f"irlosc_comm_create failed ({rc}): {self.lib.irlosc_comm_last_error(None).decode()}")
        self._h = h
        self._path = path

    def _chk(self, rc):
        if rc != 0:
            raise _lib.IrloscError(f"RCCL error {rc}: {self.lib.irlosc_comm_last_error(self._h).decode()}")

    def reduce(self, steps_done: float, elapsed_s: float) -> Tuple[float, float]:
        """-> (sum of steps over ranks, max elapsed over ranks); also a barrier."""
        s, e = C.c_double(float(steps_done)), C.c_double(float(elapsed_s))
        self._chk(self.lib.irlosc_bench_allreduce(self._h, C.byref(s), C.byref(e)))
        return s.value, e.value

    def barrier(self):
        self.reduce(0.0, 0.0)

    def allgather_u64(self, mine: int) -> List[int]:
        out = (C.c_uint64 * self.world)()
        self._chk(self.lib.irlosc_comm_allgather_u64(self._h, C.c_uint64(mine & (2 ** 64 - 1)), out))
        return [int(v) for v in out]

    def close(self):
        if getattr(self, "_h", None):
            self.lib.irlosc_comm_destroy(self._h)
            self._h = None
            if self.rank == 0:
                try:
                    os.remove(self._path)
                except OSError:
                    pass


class FileComm:
    """The same three collectives through files in TMPDIR (one node): what bench.py falls back to when RCCL cannot be
    brought up, so that a multi-GPU run still reports its line (and says so).  Every operation has a sequence number;
    a rank publishes `<base>_op<seq>_r<rank>` (atomic rename), then reads all `world` files of that operation."""

    def __init__(self, rank: int, world: int, tag: Optional[str] = None, timeout_s: float = 300.0):
        self.rank, self.world, self.timeout_s = rank, world, timeout_s
        self._base = rendezvous_path(tag) + "_file"
        self._seq = 0

    def _name(self, seq: int, r: int) -> str:
        return f"{self._base}_op{seq}_r{r}"

    def _exchange(self, payload: bytes) -> List[bytes]:
        seq = self._seq
        self._seq += 1
        tmp = self._name(seq, self.rank) + ".tmp"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, self._name(seq, self.rank))
        out, t0 = [], time.time()
        for r in range(self.world):
            while True:
                try:
                    with open(self._name(seq, r), "rb") as f:
                        data = f.read()
                    if len(data) == len(payload):
                        out.append(data)
                        break
                except OSError:
                    pass
                if time.time() - t0 > self.timeout_s:
                    raise TimeoutError(f"rank {self.rank}: rank {r} never reached operation {seq}")
                time.sleep(0.002)
        if seq >= 2:                                   # everybody has published seq - 1, hence finished reading seq - 2
            try:
                os.remove(self._name(seq - 2, self.rank))
            except OSError:
                pass
        return out

    def reduce(self, steps_done: float, elapsed_s: float) -> Tuple[float, float]:
        vals = [np.frombuffer(b, dtype=np.float64) for b in self._exchange(np.array([steps_done, elapsed_s], dtype=np.float64).tobytes())]
        return float(sum(v[0] for v in vals)), float(max(v[1] for v in vals))

    def barrier(self):
        self.reduce(0.0, 0.0)

    def allgather_u64(self, mine: int) -> List[int]:
        return [int(np.frombuffer(b, dtype=np.uint64)[0]) for b in self._exchange(np.array([mine & (2 ** 64 - 1)], dtype=np.uint64).tobytes())]

    def close(self):
        """Two closing barriers: once the second is through, everybody has finished reading the first, so everything up
        to it can go.  The (16-byte) files of the very last operation stay: no rank can know when the others have read them."""
        self.barrier()
        self.barrier()
        for seq in range(max(0, self._seq - 3), self._seq - 1):
            try:
                os.remove(self._name(seq, self.rank))
            except OSError:
                pass


def make_comm(rank: int, world: int, hip_device: int, tag: Optional[str] = None, init_timeout_s: float = 120.0):
    """-> (communicator, note).  RCCL when EVERY rank brought it up, files otherwise: the choice is made collectively --
    each rank publishes an ok / failed marker in the rendezvous directory and reads all of them -- because a job where
    some ranks barrier through RCCL and others through files never finishes.  ncclCommInitRank runs in a helper thread
    with a deadline (a rank whose peers failed early would wait in it forever); a communicator that came up on this rank
    while another rank failed is abandoned, not destroyed (its destruction may wait for the missing peers too)."""
    box = {}

    def init():
        try:
            box["comm"] = RcclComm(rank, world, hip_device, tag, rdv_timeout_s=init_timeout_s)
        except Exception as e:                                   # noqa: BLE001 - reported in the bench line
            box["err"] = str(e)
    th = threading.Thread(target=init, daemon=True)
    th.start()
    th.join(init_timeout_s)
    ok = "comm" in box
    why = box.get("err", "ncclCommInitRank did not return in %.0f s" % init_timeout_s) if not ok else ""
    d = rendezvous_dir(tag)
    tmp = os.path.join(d, f"transport_{rank}.tmp")
    with open(tmp, "wb") as f:
        f.write(b"1" if ok else b"0")
    os.replace(tmp, os.path.join(d, f"transport_{rank}"))
    votes, t0 = [], time.time()
    for r in range(world):
        while True:
            try:
                with open(os.path.join(d, f"transport_{r}"), "rb") as f:
                    v = f.read()
                if len(v) == 1:
                    votes.append(v == b"1")
                    break
            except OSError:
                pass
            if time.time() - t0 > init_timeout_s + 60.0:
                raise TimeoutError(f"rank {rank}: rank {r} never announced its transport")
            time.sleep(0.01)
    if all(votes):
        return box["comm"], "RCCL (through the C ABI)"
    failed = [r for r, v in enumerate(votes) if not v]
    return FileComm(rank, world, tag), (f"files in TMPDIR (RCCL could not be brought up on rank(s) {failed}"
                                        + (f": {why}" if why else "") + ")")


def reduce_throughput(steps_done: float, elapsed_s: float, comm=None):
    """-> (total steps over all ranks, max elapsed over ranks, whole-job steps/s).  `comm` = anything with
    reduce(steps, elapsed) -> (sum, max) (RcclComm on the GPU box); None = single process."""
    if comm is None:
        return float(steps_done), float(elapsed_s), float(steps_done) / float(elapsed_s)
    total, worst = comm.reduce(steps_done, elapsed_s)
    return float(total), float(worst), float(total) / float(worst)
