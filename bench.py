#!/usr/bin/env python3
"""bench.py -- images/sec of the ssdseglib hot path on MI355X (contract: see the task statement / DESIGN.md).

Workload at N=1 = BASELINE.json configs[1]: MobileNetV2 backbone (blocks 0..16, reference models.py:169-215)
forward + backward at batch 32, 480x640x3, fp32, with synthetic upstream gradients on the three tensors the heads
tap (`backbone-block16-project-batchnorm`, `backbone-block3-expand-relu6`, `backbone-block13-expand-relu6`), followed
by the gradient all-reduce (N>1) and the Adam update.  `--workload full` runs configs[2] (whole multi-task train step).

One process per GPU: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`; torch is used only as
plumbing (process group = RCCL over xGMI, gradient bucket tensor); all compute goes through libssdseg_hip.so.
Each rank works on its own shard of the global batch (weak scaling), one collective per step (flat fp32 gradient
bucket, sum then x 1/N inside Adam).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed inside the C library on the launch
stream) and `cpu_baseline` (NumPy oracle of the same step on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured float4-copy ceiling
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32-input MFMA dense peak

TAPS = ['backbone-block16-project-batchnorm', 'backbone-block3-expand-relu6', 'backbone-block13-expand-relu6']
IMAGE_SHAPE = (480, 640, 3)


def build_backbone_model():
    import ssdseglib
    from ssdseglib import _graph as K
    K.set_seed(1993)
    dummy = np.zeros(4, np.float32)
    b = ssdseglib.models.MobileNetV2SsdSegBuilder(IMAGE_SHAPE, 6, 4, dummy, dummy, dummy, dummy, (0.1, 0.1, 0.2, 0.2))
    inp = b._mobilenetv2_backbone()
    return K.Model(inputs=inp, outputs=[b._layers[n] for n in TAPS])


def synthetic_images(batch, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (batch,) + IMAGE_SHAPE, dtype=np.uint8).astype(np.float32)


class BackboneStep:
    """configs[1]: fwd + bwd (+ all-reduce) + Adam of the MobileNetV2 backbone on a resident batch"""

    def __init__(self, ctx, batch, rank, world, grad_bucket=None, allreduce=None):
        from ssdseglib import _engine as E
        self.model = build_backbone_model()
        self.eng = E.Engine(self.model, batch, training=True, ctx=ctx, grad_bucket=grad_bucket)
        self.ctx, self.world, self.allreduce = ctx, world, allreduce
        self.eng.set_input(synthetic_images(batch, 1993 + rank))
        rng = np.random.default_rng(7 + rank)
        self.seeds = []
        for i, t in enumerate(self.model.outputs):
            shape = (batch,) + tuple(t.shape[1:])
            g = (rng.standard_normal(shape, dtype=np.float32) * np.float32(1e-3))
            self.seeds.append(ctx.array(g))
        self.batch = batch

    def __call__(self):
        e = self.eng
        e.forward()
        for i, g in enumerate(self.seeds):
            e.seed_output_grad(i, g)
        e.backward_from_outputs()
        if self.allreduce is not None:
            self.allreduce()
        e.adam_step(lr=1e-4, grad_scale=1.0 / self.world)


def cpu_baseline_backbone(sample_batch=1):
    """NumPy oracle of the same step (fwd + bwd + Adam) on the host cores, bounded sample."""
    from oracle.np_model import NpModel
    from oracle import np_ops as O
    model = build_backbone_model()
    x = synthetic_images(sample_batch, 1993)
    ref = NpModel(model, dtype=np.float32)
    rng = np.random.default_rng(7)
    t0 = time.perf_counter()
    outs = ref.forward(x, training=True)
    gouts = [(rng.standard_normal(o.shape, dtype=np.float32) * np.float32(1e-3)) for o in outs]
    grads = ref.backward(gouts)
    for lname, gs in grads.items():
        for wname, g in gs.items():
            p = ref.weights[lname][wname]
            O.adam_step(p, g.astype(np.float32), np.zeros_like(p), np.zeros_like(p), 1)
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    return dict(value=sample_batch / dt, unit="images/sec", cores=int(threads), kind="port",
                sample=f"{sample_batch} image(s) 480x640 fwd+bwd+Adam of the same backbone through the NumPy oracle (CPU restatement of "
                       f"ssdseglib, not TensorFlow); BLAS matmuls on {threads} threads, elementwise passes on 1; {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--workload", default="backbone", choices=["backbone"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs one process per GPU: launch with "
                  f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py ...`", file=sys.stderr)
            sys.exit(2)
        args.gpus = world

    from ssdseglib import _hip as H
    dist = None
    grad_bucket = None
    allreduce = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        stream = torch.cuda.current_stream().cuda_stream      # our kernels and RCCL order on torch's current stream
        ctx = H.Context(local_rank, stream=stream)
    else:
        ctx = H.Context(local_rank)

    if world > 1:
        # the flat gradient bucket is a torch tensor (so RCCL can reduce it in place); the library sees a raw pointer
        n_params = sum(int(l.weights[w].size) for l in build_backbone_model().layers for w in l.trainable_names)
        bucket_t = torch.zeros(n_params, dtype=torch.float32, device=f"cuda:{local_rank}")
        grad_bucket = ctx.borrow(bucket_t.data_ptr(), (n_params,), np.float32, owner=bucket_t)

        def allreduce():
            dist.all_reduce(bucket_t, op=dist.ReduceOp.SUM)

    step = BackboneStep(ctx, args.batch, rank, world, grad_bucket, allreduce)

    def barrier():
        ctx.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    if not args.no_kernel_timing:
        ctx.timing(True)
        ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    if dist is not None:
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    report = ctx.timing_report() if not args.no_kernel_timing else {}
    ctx.timing(False)

    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = args.batch * world * args.steps / elapsed
        out = {
            "metric": "images/sec (fwd+bwd, 480x640)", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: MobileNetV2 backbone-only fwd+bwd(+all-reduce)+Adam, batch 32/GPU, 480x640x3, "
                                   "synthetic upstream gradients on the 3 head taps",
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "parallelism": f"dp{world}",
                       "device": ctx.device_name()},
        }
        if report:
            total_ms = sum(v["ms"] for v in report.values())
            name, dom = max(report.items(), key=lambda kv: kv[1]["ms"])
            avg_ms = dom["ms"] / dom["count"]
            gbs = dom["bytes"] / dom["count"] / (avg_ms * 1e-3) / 1e9
            tfs = dom["flops"] / dom["count"] / (avg_ms * 1e-3) / 1e12
            hbm_frac, mfma_frac = gbs / HBM_PEAK_GBS, tfs / MFMA_F32_PEAK_TFLOPS
            bound = "mfma" if mfma_frac > hbm_frac else "hbm"
            out["roofline"] = {
                "kernel": name, "bound": bound,
                "achieved": round(tfs if bound == "mfma" else gbs, 2), "peak": MFMA_F32_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS,
                "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": round(mfma_frac if bound == "mfma" else hbm_frac, 4),
                "traffic": None, "launches": dom["count"], "avg_launch_ms": round(avg_ms, 4),
                "share_of_kernel_time": round(dom["ms"] / total_ms, 4),
                "algorithmic_bytes_per_launch": dom["bytes"] / dom["count"], "flops_per_launch": dom["flops"] / dom["count"],
            }
            top = sorted(report.items(), key=lambda kv: -kv[1]["ms"])[:12]
            out["kernels"] = [
                {"kernel": k, "launches": v["count"], "ms_per_step": round(v["ms"] / args.steps, 4),
                 "GB/s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0.0,
                 "TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0.0} for k, v in top]
            out["kernel_ms_per_step"] = round(total_ms / args.steps, 3)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_backbone(1)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
