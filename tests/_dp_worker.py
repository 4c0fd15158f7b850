"""One rank of the two-process data-parallel GPU test (tests/test_gpu_data_parallel.py): a FRESH process that runs the real
Engine on cuda:0 for its shard of the global batch, reduces through the communicator named by SSDSEG_COMM and dumps the
result.  usage: python tests/_dp_worker.py <out_dir>   (RANK / WORLD_SIZE / SSDSEG_COMM / SSDSEG_COMM_DIR from the environment)"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))


def main():
    out_dir = sys.argv[1]
    from ssdseglib import _engine as E, _hip as H, _parallel as P
    from tests.test_gpu_data_parallel import build_case, GLOBAL_BATCH
    rank, _, world = P.env_world()
    ctx = H.Context(0)
    comm = P.init_comm(ctx)
    model, x, targets = build_case()
    lo, hi = P.shard_bounds(GLOBAL_BATCH, rank, world)
    eng = E.Engine(model, hi - lo, training=True, ctx=ctx)
    eng.configure_losses(model._compiled["loss"], model._compiled["loss_weights"])
    reducer = P.GradientAllReduce(comm, eng)
    eng.train_step(x[lo:hi], {k: v[lo:hi] for k, v in targets.items()}, optimizer=model._compiled["optimizer"], allreduce=reducer, world=world)
    ctx.sync()
    drift = reducer.check_replicas_in_sync()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), grads=eng.P["grads"].download(), params=eng.P["params"].download(),
             state=eng.P["state"].download(), drift=np.float64(drift))
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
