"""Data parallelism over the 8 GPUs of a node: one process per GPU, replicated weights, ONE collective per step.

The reference has no distributed code; this is the new part.  Work shards by image: global batch
-> `world` equal contiguous shards, every rank runs the full step on its shard (BatchNorm statistics and the hard-negative
mining pool are per-replica, exactly what `tf.distribute.MirroredStrategy` would give the reference), then the flat fp32
gradient bucket (4,009,920 floats = 16 MB) is summed across ranks -- together with the mean of the BatchNormalization moving
statistics (37,488 floats) in the same RCCL group -- and Adam applies it scaled by 1/world, identically on every rank, so
weights never need a broadcast after step 0.  The result equals the mean of `world` independent reference steps taken
from the same weights.

No torch here: the collective is RCCL behind the C-ABI (`ssdseg_comm_*`, `ssdseg_allreduce_grads`, csrc/comm.hip).  The only
host-side exchange is the 128-byte RCCL unique id, passed through a file handshake (`exchange_bytes`: fresh random bytes
from both sides, so a stale file of a crashed launch is never taken for this one's) at a path keyed by the launcher's
environment: `bench.py --gpus N` starts its own N rank processes and names the path (SSDSEG_RDZV_FILE); under
`python -m torch.distributed.run` (which only provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT) it is derived from those.  xGMI is
point-to-point (7 links x ~153 GB/s), a 16 MB ring all-reduce is ~0.2 ms against a >= 25 ms step, so a single flat bucket
is the right granularity (no bucketing / overlap machinery is worth its launches here).

`HostStagedComm` is a REHEARSAL transport, selected only explicitly (tests; SSDSEG_COMM=host): RCCL refuses two ranks on one
device, so the two-process test on a one-GPU box sums the buckets through host memory in rank order instead.  It is never
chosen implicitly and bench.py never uses it on a multi-GPU node.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import tempfile
import time
from typing import Optional, Tuple

import numpy as np

from . import _hip as H

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL needs it)

ID_BYTES = 128
COMM_F32, COMM_F64 = 0, 1
COMM_SUM, COMM_MAX = 0, 1


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the launcher's environment (defaults: single process)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(global_batch: int, rank: int, world: int, allow_uneven: bool = False) -> Tuple[int, int]:
    """contiguous shard [begin, end) of `rank`.  Shards must be equal: every rank contributes its shard MEAN and the collective
    averages those with equal weights, so an uneven split would turn the result into a mean of means.  `allow_uneven` is for
    callers that only partition work (no gradient averaging); sizes then differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(global_batch, world)
    if extra and not allow_uneven:
        raise ValueError(f"global batch {global_batch} does not split evenly over {world} ranks: gradient averaging with equal "
                         f"weights would become a mean of means")
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


# ---------------------------------------------------------------------------------------------- rendezvous (128 bytes, one node)
def rendezvous_path() -> str:
    """file that carries rank 0's RCCL unique id to the other ranks of THIS launch.  SSDSEG_RDZV_FILE overrides; otherwise the
    name is built from what all workers of one `torch.distributed.run` launch share and other launches do not: MASTER_PORT, the
    launcher's pid (every worker is its child) and the elastic restart count."""
    explicit = os.environ.get("SSDSEG_RDZV_FILE")
    if explicit:
        return explicit
    key = "_".join([os.environ.get("MASTER_PORT", "0"), str(os.getppid()), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"),
                    os.environ.get("TORCHELASTIC_RUN_ID", "none")])
    return os.path.join(tempfile.gettempdir(), f"ssdseg_rdzv_{key}")


NONCE_BYTES = 16


def _publish(path: str, data: bytes) -> None:
    """atomic: write to a private name, fsync, rename"""
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "wb") as f:
        f.write(data)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def _read(path: str) -> bytes:
    try:
        with open(path, "rb") as f:
            return f.read()
    except OSError:
        return b""


def rendezvous_cleanup(path: str) -> None:
    """remove the files of a (finished or crashed) rendezvous on `path`"""
    import glob
    for p in [path, path + ".go"] + glob.glob(glob.escape(path) + ".ack*"):
        try:
            os.unlink(p)
        except OSError:
            pass


def exchange_bytes(rank: int, payload: Optional[bytes], nbytes: int, path: Optional[str] = None, timeout_s: float = 300.0,
                   world: int = 2) -> bytes:
    """rank 0's `payload` (nbytes) reaches every other rank of THIS launch, through three kinds of small files next to `path`.

    A bare "rank 0 writes, the others read whatever is there" is not safe: a crashed earlier launch with the same path leaves a
    stale id behind, a later rank > 0 reads it before rank 0 republishes and then blocks in ncclCommInitRank for ever.  So the
    exchange is a handshake in which every side proves freshness with random bytes made in THIS process:
      rank 0: removes whatever is at `path` / `path.go` / `path.ack*`, publishes `path` = nonce | payload, waits until every
              rank k has published `path.ack<k>` = nonce | token_k for THIS nonce, then publishes `path.go` = nonce | token_1 |
              ... | token_{world-1};
      rank k: draws token_k once; acknowledges every new nonce it sees at `path`; returns the payload read together with the
              acknowledged nonce only when `path.go` names that nonce AND echoes token_k -- a stale `path` or `path.go` can do
              neither.
    No clocks involved.  TimeoutError (with the path and what was missing) after `timeout_s`; callers exit non-zero on it."""
    path = path or rendezvous_path()
    deadline = time.monotonic() + timeout_s
    if rank == 0:
        assert payload is not None and len(payload) == nbytes
        rendezvous_cleanup(path)
        nonce = os.urandom(NONCE_BYTES)
        _publish(path, nonce + payload)
        tokens = {}
        while len(tokens) < world - 1:
            for k in range(1, world):
                if k not in tokens:
                    ack = _read(f"{path}.ack{k}")
                    if len(ack) == 2 * NONCE_BYTES and ack[:NONCE_BYTES] == nonce:
                        tokens[k] = ack[NONCE_BYTES:]
            if len(tokens) < world - 1:
                if time.monotonic() > deadline:
                    missing = [k for k in range(1, world) if k not in tokens]
                    raise TimeoutError(f"rank 0: ranks {missing} never acknowledged the rendezvous at {path} within {timeout_s:.0f} s "
                                       f"(did they start, and with the same SSDSEG_RDZV_FILE / launcher?)")
                time.sleep(0.005)
        _publish(path + ".go", nonce + b"".join(tokens[k] for k in range(1, world)))
        return payload
    token = os.urandom(NONCE_BYTES)
    acked, current = None, None
    while True:
        data = _read(path)
        if len(data) == NONCE_BYTES + nbytes and data[:NONCE_BYTES] != acked:
            acked, current = data[:NONCE_BYTES], data[NONCE_BYTES:]
            _publish(f"{path}.ack{rank}", acked + token)
        if acked is not None:
            go = _read(path + ".go")
            lo = NONCE_BYTES * rank
            if len(go) == NONCE_BYTES * world and go[:NONCE_BYTES] == acked and go[lo:lo + NONCE_BYTES] == token:
                return current
        if time.monotonic() > deadline:
            raise TimeoutError(f"rank {rank}: rendezvous at {path} not completed within {timeout_s:.0f} s "
                               f"({'no file from rank 0' if acked is None else 'rank 0 never confirmed (stale file of an earlier launch?)'})")
        time.sleep(0.005)


class RcclComm:
    """the RCCL communicator of one rank, bound to its HIP context (csrc/comm.hip)"""

    transport = "rccl"

    def __init__(self, ctx: H.Context, rank: int, world: int, path: Optional[str] = None):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self._scratch = None
        lib = ctx.lib
        ident = None
        if self.rank == 0:
            buf = C.create_string_buffer(ID_BYTES)
            H._check(lib.ssdseg_comm_unique_id(buf, ID_BYTES), "ssdseg_comm_unique_id")
            ident = buf.raw
        path = path or rendezvous_path()
        timeout = float(os.environ.get("SSDSEG_RDZV_TIMEOUT", "300"))
        ident = exchange_bytes(self.rank, ident, ID_BYTES, path, timeout, self.world) if self.world > 1 else ident
        H._check(lib.ssdseg_comm_init_rank(ctx.handle, ident, ID_BYTES, self.rank, self.world), "ssdseg_comm_init_rank")
        if self.rank == 0 and self.world > 1:
            rendezvous_cleanup(path)        # init_rank is collective: once it has returned here every rank holds the id

    def group_size(self) -> int:
        """the world size the RCCL communicator itself reports (ssdseg_comm_info)"""
        r, w = C.c_int(-1), C.c_int(-1)
        H._check(self.ctx.lib.ssdseg_comm_info(self.ctx.handle, C.byref(r), C.byref(w)), "ssdseg_comm_info")
        return int(w.value)

    def allreduce_grads(self, grads: H.DeviceBuffer, state: Optional[H.DeviceBuffer] = None):
        """in place: grads <- SUM over ranks, state <- MEAN over ranks; stream-ordered, no host sync"""
        self.ctx.call("ssdseg_allreduce_grads", grads, C.c_size_t(grads.size), state, C.c_size_t(state.size if state is not None else 0))

    def _reduce_scalar(self, value: float, op: int) -> float:
        if self._scratch is None:
            self._scratch = self.ctx.empty(1, np.float64)
        self._scratch.upload(np.array([value], np.float64))
        self.ctx.call("ssdseg_allreduce", self._scratch, C.c_size_t(1), COMM_F64, op)
        return float(self._scratch.download()[0])

    def max(self, value: float) -> float:
        return self._reduce_scalar(value, COMM_MAX)

    def sum(self, value: float) -> float:
        return self._reduce_scalar(value, COMM_SUM)

    def barrier(self):
        self.ctx.sync()
        self._reduce_scalar(0.0, COMM_SUM)     # the download inside blocks until every rank has contributed

    def broadcast(self, buf: H.DeviceBuffer, root: int = 0):
        assert buf.dtype == np.float32
        self.ctx.call("ssdseg_broadcast", buf, C.c_size_t(buf.size), int(root))

    def close(self):
        if self.ctx.handle:
            H._check(self.ctx.lib.ssdseg_comm_destroy(self.ctx.handle), "ssdseg_comm_destroy")


class HostStagedComm:
    """REHEARSAL transport for ranks that share one device (RCCL rejects duplicate devices) or have none: buffers are staged
    through files of a shared directory and summed on the host in rank order.  Same interface as RcclComm; explicit opt-in only
    (tests, SSDSEG_COMM=host).  `ctx` may be None (host arrays only: the CPU tests)."""

    transport = "host"

    def __init__(self, ctx: Optional[H.Context], rank: int, world: int, directory: str):
        self.ctx, self.rank, self.world, self.dir = ctx, int(rank), int(world), directory
        self.round = 0
        os.makedirs(directory, exist_ok=True)

    def _exchange(self, arr: np.ndarray) -> list:
        """everybody publishes `arr`, returns the list of all ranks' arrays in rank order"""
        self.round += 1
        mine = os.path.join(self.dir, f"r{self.round}_{self.rank}.npy")
        tmp = mine + ".tmp"
        with open(tmp, "wb") as f:
            np.save(f, arr)
        os.replace(tmp, mine)
        out = []
        deadline = time.monotonic() + 300.0
        for r in range(self.world):
            p = os.path.join(self.dir, f"r{self.round}_{r}.npy")
            while not os.path.exists(p):
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {self.rank}: rank {r} never published round {self.round}")
                time.sleep(0.002)
            out.append(np.load(p))
        # round k-2 is read by everyone once all ranks have published round k-1 ... keep two rounds, drop the older ones
        old = os.path.join(self.dir, f"r{self.round - 2}_{self.rank}.npy")
        if self.round > 2 and os.path.exists(old):
            os.unlink(old)
        return out

    def group_size(self) -> int:
        """number of ranks that actually took part in an exchange (counted, not configured)"""
        return len(self._exchange(np.zeros(1)))

    def allreduce_array(self, arr: np.ndarray, mean: bool = False) -> np.ndarray:
        parts = self._exchange(np.ascontiguousarray(arr))
        total = parts[0].copy()
        for p in parts[1:]:
            total += p                       # fixed rank order: every rank computes bit-identical sums
        return total / self.world if mean else total

    def allreduce_grads(self, grads, state=None):
        if isinstance(grads, np.ndarray):
            grads[...] = self.allreduce_array(grads)
            if state is not None:
                state[...] = self.allreduce_array(state, mean=True)
            return
        self.ctx.join()
        grads.upload(self.allreduce_array(grads.download()))
        if state is not None and state.size:
            state.upload(self.allreduce_array(state.download(), mean=True).astype(np.float32))

    def max(self, value: float) -> float:
        return float(max(float(p[0]) for p in self._exchange(np.array([value], np.float64))))

    def sum(self, value: float) -> float:
        return float(sum(float(p[0]) for p in self._exchange(np.array([value], np.float64))))

    def barrier(self):
        if self.ctx is not None:
            self.ctx.sync()
        self._exchange(np.zeros(1))

    def broadcast(self, buf, root: int = 0):
        parts = self._exchange(buf.download() if not isinstance(buf, np.ndarray) else buf)
        if isinstance(buf, np.ndarray):
            buf[...] = parts[root]
        else:
            buf.upload(parts[root])

    def close(self):
        pass


def init_comm(ctx: Optional[H.Context], transport: Optional[str] = None):
    """communicator for this process from the launcher's environment; None for a single process.  transport: "rccl" (default)
    or "host" (explicit rehearsal: needs SSDSEG_COMM_DIR, a directory shared by the ranks)."""
    rank, _, world = env_world()
    if world == 1:
        return None
    transport = transport or os.environ.get("SSDSEG_COMM", "rccl")
    if transport == "rccl":
        return RcclComm(ctx, rank, world)
    if transport == "host":
        directory = os.environ.get("SSDSEG_COMM_DIR")
        if not directory:
            raise ValueError("SSDSEG_COMM=host needs SSDSEG_COMM_DIR (a directory all ranks share)")
        return HostStagedComm(ctx, rank, world, directory)
    raise ValueError(f"unknown SSDSEG_COMM transport {transport!r} (rccl | host)")


class GradientAllReduce:
    """The step's one collective for an `Engine`: sums its flat gradient bucket and averages its BatchNormalization moving
    statistics across the ranks.  `__call__()` is enqueued on the context's stream (after joining the weight-gradient side
    stream), so it is ordered after the backward kernels and before Adam without host synchronisation; Adam then multiplies
    the gradients by `scale` = 1/world."""

    def __init__(self, comm, engine):
        self.comm, self.engine = comm, engine
        self.world = comm.world if comm is not None else 1
        self.scale = 1.0 / self.world

    def __call__(self):
        if self.world > 1:
            P = self.engine.P
            self.comm.allreduce_grads(P["grads"], P["state"] if P["n_st"] else None)

    def check_replicas_in_sync(self) -> float:
        """max |p - p_rank0| over the group (debug aid: replicas must stay bit-identical)"""
        if self.world == 1:
            return 0.0
        p = self.engine.P["params"]
        mine = p.download()
        ref = self.engine.ctx.array(mine)
        self.comm.broadcast(ref, 0)
        return self.comm.max(float(np.abs(ref.download() - mine).max()))
