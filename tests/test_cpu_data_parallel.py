"""world_size-2 data-parallel path on CPU: batch sharding, flat-bucket all-reduce, 1/world scaling, replicas in sync -- and
the parity statement of DESIGN.md: a DP step == the mean of `world` independent steps from the same weights (checked with
the NumPy oracle standing in for the per-rank HIP step).  Two transports: torch.distributed gloo (test-local plumbing) and the
product package's explicit host-staged rehearsal transport; plus the 128-byte file rendezvous the RCCL path uses."""
import multiprocessing
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds():
    from ssdseglib._parallel import shard_bounds
    assert [shard_bounds(256, r, 8) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]
    with pytest.raises(ValueError, match="mean of means"):          # equal weights in the collective need equal shards
        shard_bounds(19, 0, 4)
    parts = [shard_bounds(19, r, 4, allow_uneven=True) for r in range(4)]
    assert parts == [(0, 5), (5, 10), (10, 15), (15, 19)]
    with pytest.raises(ValueError):
        shard_bounds(8, 4, 4)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tiny_model():
    from ssdseglib import _graph as K
    K.set_seed(21)
    inp = K.Input((16, 16, 3), name='backbone-input')
    x = K.Rescaling(1 / 127.5, -1, name='rescale')(inp)
    x = K.Conv2D(8, 3, strides=2, use_bias=False, name='stem')(x)
    x = K.BatchNormalization(name='stem-bn')(x)
    x = K.ReLU(max_value=6.0, name='stem-relu')(x)
    x = K.DepthwiseConv2D(3, use_bias=False, name='dw')(x)
    x = K.BatchNormalization(name='dw-bn')(x)
    x = K.ReLU(max_value=6.0, name='dw-relu')(x)
    x = K.Conv2D(4, 1, use_bias=False, name='pw')(x)
    x = K.BatchNormalization(name='pw-bn')(x)
    return K.Model(inp, x)


def _flat(model, names_attr):
    return np.concatenate([l.weights[w].reshape(-1) for l in model.layers for w in l.weights if (w in l.trainable_names) == (names_attr == "trainable")])


def _shard_step(model, x, seed):
    """one oracle step on a shard -> (flat gradient bucket, flat moving-statistics bucket after the step), engine order"""
    from oracle.np_model import NpModel
    ref = NpModel(model, dtype=np.float64)
    (out,) = ref.forward(x, training=True)
    g = np.random.default_rng(seed).normal(size=out.shape)
    grads = ref.backward([g / x.shape[0]])
    flat_g = np.concatenate([grads[l.name][w].reshape(-1) for l in model.layers for w in l.trainable_names])
    flat_s = np.concatenate([np.asarray(ref.weights[l.name][w], np.float64).reshape(-1) for l in model.layers for w in l.weights
                             if w not in l.trainable_names])
    # (the oracle's forward pass leaves the moving statistics alone; a shard-dependent offset stands in for each replica's update)
    flat_s = flat_s + 0.01 * np.random.default_rng(seed + 1000).normal(size=flat_s.shape)
    return flat_g, flat_s


def _setup_paths():
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))


def _gloo_worker(rank, world, port, global_batch, out_dir):
    _setup_paths()
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import np_ops as O
    from ssdseglib import _parallel as P
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert P.env_world() == (rank, rank, world)
    model = _tiny_model()
    x = np.random.default_rng(0).integers(0, 256, (global_batch, 16, 16, 3)).astype(np.float64)
    lo, hi = P.shard_bounds(global_batch, rank, world)
    g, st = _shard_step(model, x[lo:hi], seed=100 + rank)
    bucket, state = torch.from_numpy(g.copy()), torch.from_numpy(st.copy())
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM)                  # the one collective of the step: gradients summed ...
    dist.all_reduce(state, op=dist.ReduceOp.SUM)                   # ... moving statistics averaged
    state /= world
    params = _flat(model, "trainable").astype(np.float64)
    new, _, _ = O.adam_step(params, bucket.numpy() / world, np.zeros_like(params), np.zeros_like(params), 1, lr=1e-2)
    ref = torch.from_numpy(new.copy())
    dist.broadcast(ref, src=0)
    assert float((ref - torch.from_numpy(new)).abs().max()) == 0.0   # replicas in sync
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.concatenate([new, bucket.numpy() / world, state.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def _expected(world, global_batch):
    _setup_paths()
    from ssdseglib._parallel import shard_bounds
    model = _tiny_model()
    x = np.random.default_rng(0).integers(0, 256, (global_batch, 16, 16, 3)).astype(np.float64)
    shards = [shard_bounds(global_batch, r, world) for r in range(world)]
    steps = [_shard_step(model, x[lo:hi], seed=100 + r) for r, (lo, hi) in enumerate(shards)]
    return np.mean([s[0] for s in steps], axis=0), np.mean([s[1] for s in steps], axis=0)


def test_two_rank_gloo_step_equals_mean_of_independent_steps(tmp_path):
    world, global_batch = 2, 6
    port = _free_port()
    mp.spawn(_gloo_worker, args=(world, port, global_batch, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1), "replicas diverged"
    mean_g, mean_s = _expected(world, global_batch)
    n = mean_g.size
    assert np.allclose(r0[n:2 * n], mean_g, rtol=0, atol=1e-14)
    assert np.allclose(r0[2 * n:], mean_s, rtol=0, atol=1e-14)


def _host_worker(rank, world, global_batch, comm_dir, out_dir, rdzv):
    _setup_paths()
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), SSDSEG_COMM="host", SSDSEG_COMM_DIR=comm_dir)
    from ssdseglib import _parallel as P
    # the 128-byte rendezvous the RCCL path uses (rank 0 publishes, the others poll)
    ident = bytes(range(128)) if rank == 0 else None
    got = P.exchange_bytes(rank, ident, 128, rdzv, timeout_s=60, world=world)
    assert got == bytes(range(128))
    comm = P.init_comm(None)
    assert comm.transport == "host" and comm.world == world
    model = _tiny_model()
    x = np.random.default_rng(0).integers(0, 256, (global_batch, 16, 16, 3)).astype(np.float64)
    lo, hi = P.shard_bounds(global_batch, rank, world)
    g, st = _shard_step(model, x[lo:hi], seed=100 + rank)
    comm.allreduce_grads(g, st)                                    # in place: g <- sum, st <- mean
    assert comm.max(float(rank)) == world - 1 and comm.sum(1.0) == world
    comm.barrier()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.concatenate([g / world, st]))


def test_two_rank_host_staged_transport_and_rendezvous(tmp_path):
    world, global_batch = 2, 6
    comm_dir, out_dir = tmp_path / "comm", tmp_path / "out"
    out_dir.mkdir()
    # a crashed earlier launch left its id, its go file and an acknowledgement behind at the same path: nobody may take them
    (tmp_path / "rdzv").write_bytes(b"S" * 16 + b"\xee" * 128)
    (tmp_path / "rdzv.go").write_bytes(b"S" * 16 + b"T" * 16)
    (tmp_path / "rdzv.ack1").write_bytes(b"S" * 16 + b"T" * 16)
    mpctx = multiprocessing.get_context("spawn")
    procs = [mpctx.Process(target=_host_worker, args=(r, world, global_batch, str(comm_dir), str(out_dir), str(tmp_path / "rdzv")))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    r0, r1 = np.load(out_dir / "rank0.npy"), np.load(out_dir / "rank1.npy")
    assert np.array_equal(r0, r1), "replicas diverged"
    mean_g, mean_s = _expected(world, global_batch)
    assert np.allclose(r0[:mean_g.size], mean_g, rtol=0, atol=1e-14)
    assert np.allclose(r0[mean_g.size:], mean_s, rtol=0, atol=1e-14)


def test_rendezvous_handshake_ignores_stale_files_and_times_out(tmp_path):
    """the 128-byte id exchange: three ranks as threads; ranks > 0 start FIRST and find a complete stale exchange
    (id + go + their own old acknowledgement) at the path; they must come back with rank 0's fresh payload, not the stale one.
    Without a rank 0 the others time out with a message instead of returning the stale id."""
    import threading
    from ssdseglib import _parallel as P
    path = str(tmp_path / "rdzv")
    stale = b"N" * P.NONCE_BYTES

    def seed_stale():
        (tmp_path / "rdzv").write_bytes(stale + b"\x55" * 128)
        (tmp_path / "rdzv.go").write_bytes(stale + b"A" * 16 + b"B" * 16)
        (tmp_path / "rdzv.ack1").write_bytes(stale + b"A" * 16)
        (tmp_path / "rdzv.ack2").write_bytes(stale + b"B" * 16)

    seed_stale()
    fresh = bytes(range(128))
    got = {}

    def run(rank, delay):
        import time as _t
        _t.sleep(delay)
        try:
            got[rank] = P.exchange_bytes(rank, fresh if rank == 0 else None, 128, path, timeout_s=30, world=3)
        except Exception as e:        # noqa: BLE001 - reported through the assertion below
            got[rank] = e

    threads = [threading.Thread(target=run, args=(r, 0.0 if r else 0.3)) for r in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert got == {0: fresh, 1: fresh, 2: fresh}, got
    P.rendezvous_cleanup(path)
    assert not list(tmp_path.glob("rdzv*"))

    seed_stale()                       # rank 0 never shows up: the stale files alone do not complete a handshake
    with pytest.raises(TimeoutError, match="stale file"):
        P.exchange_bytes(1, None, 128, path, timeout_s=0.5, world=3)
    with pytest.raises(TimeoutError, match="never acknowledged"):
        P.exchange_bytes(0, fresh, 128, str(tmp_path / "other"), timeout_s=0.3, world=2)


def test_unknown_transport_and_missing_dir_are_errors(monkeypatch):
    from ssdseglib import _parallel as P
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("SSDSEG_COMM", "host")
    monkeypatch.delenv("SSDSEG_COMM_DIR", raising=False)
    with pytest.raises(ValueError, match="SSDSEG_COMM_DIR"):
        P.init_comm(None)
    monkeypatch.setenv("SSDSEG_COMM", "smoke-signals")
    with pytest.raises(ValueError, match="unknown"):
        P.init_comm(None)


def test_bench_launcher_without_a_gpu_fails_loudly_and_cleans_up(tmp_path):
    """`python bench.py --gpus 2` here (no GPU): the parent launches two rank processes without touching HIP itself, both die at
    Context creation (no CPU fallback), the launcher reports the first failure, stops the rest and returns non-zero -- no hang,
    no JSON line, no leftover rendezvous directory."""
    import subprocess
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SSDSEG_RDZV_FILE")}
    env["TMPDIR"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = r.stderr.decode(errors="replace")
    assert r.returncode != 0 and "exited with" in err, err[-2000:]
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert not list(tmp_path.glob("ssdseg_bench_*"))
    # a launcher-provided world that disagrees with --gpus is an argument error, not a silent single-rank run
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE is 2" in r.stderr.decode()
