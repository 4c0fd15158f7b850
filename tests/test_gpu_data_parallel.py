"""Data parallelism on the GPU box (one card): (1) the RCCL entry points of the C-ABI with a one-rank communicator -- unique id,
ncclCommInitRank, the grouped all-reduce of gradient + state buckets, max / broadcast -- stream-ordered against real kernels;
(2) two FRESH child processes, each running the real Engine on cuda:0 for its shard, reduced through the explicit host-staged
rehearsal transport (RCCL refuses two ranks on one device): the data-parallel step equals the mean of two single steps taken
from the same weights -- gradients bit for bit, parameters after Adam, averaged BatchNormalization moving statistics.
No RCCL scaling curve exists from this box; the 8-GPU run is the driver's."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import np_ops as O

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLOBAL_BATCH = 6


def build_case():
    """the 96x128 multi-task model of test_gpu_full_model with deterministic weights, a global batch and its targets"""
    import ssdseglib
    from tests.test_gpu_full_model import CW, SHAPE, build, make_targets
    rng = np.random.default_rng(77)
    boxes, builder, model = build(seed=5)
    for l in model.layers:
        if type(l).__name__ == "BatchNormalization":
            c = l.weights["gamma"].size
            l.weights["gamma"] = rng.uniform(0.7, 1.3, c).astype(np.float32)
            l.weights["beta"] = rng.normal(0, 0.3, c).astype(np.float32)
    _, _, targets = make_targets(rng, boxes, GLOBAL_BATCH)
    model.compile(optimizer=ssdseglib.optimizers.Adam(learning_rate=1e-3),
                  loss={'output-mask': ssdseglib.losses.cross_entropy(classes_weights=CW), 'output-labels': ssdseglib.losses.confidence_loss,
                        'output-boxes': ssdseglib.losses.localization_loss},
                  loss_weights={'output-mask': 1.0, 'output-labels': 1.0, 'output-boxes': 1.0})
    x = rng.integers(0, 256, (GLOBAL_BATCH,) + SHAPE).astype(np.float32)
    return model, x, targets


def test_rccl_one_rank_communicator_through_the_cabi(tmp_path):
    from ssdseglib import _hip as H, _parallel as P
    ctx = H.Context(0)      # own context: a communicator binds to it
    try:
        comm = P.RcclComm(ctx, 0, 1, path=str(tmp_path / "rdzv"))
        rng = np.random.default_rng(0)
        g = rng.normal(size=4_009_920).astype(np.float32)
        s = rng.normal(size=37_488).astype(np.float32)
        dg, ds = ctx.array(g), ctx.array(s)
        ctx.call("ssdseg_axpby", dg, 4, dg, 4, g.size // 4, 4, 1.0, 1.0)      # a kernel in front: dg <- 2g, the collective is ordered behind it
        comm.allreduce_grads(dg, ds)
        assert np.array_equal(dg.download(), g + g) and np.array_equal(ds.download(), s)     # sum over one rank, mean over one rank
        assert comm.max(3.5) == 3.5 and comm.sum(2.0) == 2.0
        comm.broadcast(dg, 0)
        comm.barrier()
        import ctypes as C
        r, w = C.c_int(-1), C.c_int(-1)
        assert ctx.lib.ssdseg_comm_info(ctx.handle, C.byref(r), C.byref(w)) == 0 and (r.value, w.value) == (0, 1)
        comm.close()
        with pytest.raises(H.SsdsegError):
            ctx.call("ssdseg_allreduce_grads", dg, C.c_size_t(4), None, C.c_size_t(0))     # no communicator any more: loud
    finally:
        ctx.sync()
        ctx.close()


def test_two_process_step_equals_mean_of_two_single_steps(ctx, tmp_path):
    from ssdseglib import _engine as E, _parallel as P
    world = 2
    comm_dir, out_dir = tmp_path / "comm", tmp_path / "out"
    out_dir.mkdir()
    env = dict(os.environ, WORLD_SIZE=str(world), SSDSEG_COMM="host", SSDSEG_COMM_DIR=str(comm_dir), PYTHONPATH=REPO)
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "_dp_worker.py"), str(out_dir)],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r0, r1 = np.load(out_dir / "rank0.npz"), np.load(out_dir / "rank1.npz")
    for k in ("grads", "params", "state"):
        assert np.array_equal(r0[k], r1[k]), f"replicas diverged in {k}"
    assert float(r0["drift"]) == 0.0 and float(r1["drift"]) == 0.0

    # the same two shards as single steps in THIS process, from the same initial weights
    model, x, targets = build_case()
    p0 = None
    grads, states = [], []
    for r in range(world):
        lo, hi = P.shard_bounds(GLOBAL_BATCH, r, world)
        eng = E.Engine(model, hi - lo, training=True, ctx=ctx) if r == 0 else eng
        if r == 0:
            eng.configure_losses(model._compiled["loss"], model._compiled["loss_weights"])
            p0, s0 = eng.P["params"].download(), eng.P["state"].download()
        else:
            eng.P["params"].upload(p0); eng.P["state"].upload(s0)
        eng.set_input(x[lo:hi])
        eng.set_targets({k: v[lo:hi] for k, v in targets.items()})
        eng.forward()
        eng.backward()
        ctx.sync()
        grads.append(eng.P["grads"].download())
        states.append(eng.P["state"].download())
    assert np.array_equal(r0["grads"], grads[0] + grads[1]), "all-reduced bucket != sum of the two single-step gradients (bit for bit)"
    assert np.array_equal(r0["state"], ((states[0] + states[1]) / np.float32(2)).astype(np.float32)) or \
        np.abs(r0["state"] - (states[0].astype(np.float64) + states[1]) / 2).max() < 1e-6
    mean = ((grads[0] + grads[1]) * np.float32(0.5)).astype(np.float32)
    want, _, _ = O.adam_step(p0.astype(np.float32), mean, np.zeros_like(p0), np.zeros_like(p0), 1, lr=1e-3)
    assert np.abs(r0["params"] - want).max() < 2e-6
    assert not np.array_equal(r0["params"], p0)


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it starts its own N fresh rank processes and prints
    rank 0's line.  Rehearsed on the one card of this box: both ranks on device 0, explicit host-staged transport (RCCL refuses
    two ranks on one device) -- the launch, rendezvous directory, barriers, max-over-ranks timing and the collective's place in
    the step are the ones the 8-GPU run uses; only the transport differs, and the line says so."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SSDSEG_RDZV_FILE", "SSDSEG_COMM_DIR")}
    env.update(SSDSEG_BENCH_DEVICE="0", SSDSEG_COMM="host")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-4000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    cfg = out["config"]
    assert cfg["collective"] == "host" and cfg["comm_world"] == 2 and cfg["launcher"] == "bench.py"
    assert cfg["global_batch"] == 8 and cfg["parallelism"] == "dp2"
    assert np.isfinite(out["value"]) and out["value"] > 0 and np.isfinite(out["ms_per_step"])
    assert "rccl_allreduce_grads" not in json.dumps(out)        # the host transport never claims to be RCCL
