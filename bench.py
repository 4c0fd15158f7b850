#!/usr/bin/env python3
"""bench.py -- images/sec of the ssdseglib hot path on MI355X (contract: task statement / DESIGN.md "Measurement").

Workloads (BASELINE.json `configs`):
  full (default, configs[2]; configs[3] when launched on N GPUs): the whole MobileNetV2-SSDLite-DeepLabV3+ train step at
      batch 32/GPU, 480x640x3, fp32: anchor encode -> forward -> 3 losses (weighted CE, confidence with hard-negative
      mining, localization) -> backward -> all-reduce (N > 1) -> Adam.  The largest single-GPU configuration of BASELINE.json
      and the one its metric ("images/sec fwd+bwd at 1/2/4/8 GPUs") is quoted on.
  backbone (configs[1]): MobileNetV2 backbone blocks 0..16 (reference models.py:169-215) forward + backward, synthetic
      upstream gradients on the three tensors the heads tap, all-reduce (N > 1), Adam.
  shufflenet (configs[4], per-GPU share): the full train step on the ShuffleNetV2-1x variant.

One process per GPU.  `python bench.py --gpus N` starts its own N rank processes (launch_ranks: the parent touches no GPU);
under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (which provides RANK / LOCAL_RANK / WORLD_SIZE
/ MASTER_*) the ranks are the launcher's.  Either way the collective is RCCL behind our C-ABI (ssdseg_allreduce_grads), the
128-byte RCCL id travels through a file handshake (ssdseglib/_parallel.py) -- no torch in this file or in the product path.  Weak scaling: each rank owns a
32-image shard, one collective per step (gradients summed, BatchNorm moving statistics averaged).

Rank 0 prints ONE JSON line with `roofline` (dominant kernel symbol; durations from HIP events recorded around every launch of
that kernel in the timed region, on the launch stream, inside the C library; bytes / flops per SURVEY.md 8(d)) and
`cpu_baseline` (NumPy oracle of the same step, bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is the measured float4-copy ceiling
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32-input MFMA dense peak

TAPS = ['backbone-block16-project-batchnorm', 'backbone-block3-expand-relu6', 'backbone-block13-expand-relu6']
IMAGE_SHAPE = (480, 640, 3)
CLASS_WEIGHTS = (0.05, 0.575, 0.135, 0.24)     # NB03#cell10
STDS = (0.1, 0.1, 0.2, 0.2)


def default_boxes():
    import ssdseglib
    b = ssdseglib.boxes.DefaultBoundingBoxes(feature_maps_shapes=((30, 40), (15, 20), (8, 10), (4, 5)),
                                             centers_padding_from_borders_percentage=(0.025, 0.05, 0.075, 0.1), boxes_scales=(0.15, 0.95),
                                             additional_square_box=True)                     # NB03#cell6
    b.rescale_boxes_coordinates(image_shape=IMAGE_SHAPE[:2])
    return b


def build_models(seed=1993, kind="mobilenetv2"):
    """-> (boxes, builder) for the NB03 configuration; kind "shufflenetv2" = SURVEY.md 8(d) config 5: model_size '1x', additional
    depthwise convolution, residual connections; the reference's quirk Q1 -- heads built with ReLU(max_value=0.0) -- kept
    bug-compatible.  ('1x' splits its stage-2 tensors into 58-channel branches; the engine runs those units on zero-padded
    60-channel tensors and weights.)"""
    import ssdseglib
    from ssdseglib import _graph as K
    K.set_seed(seed)
    boxes = default_boxes()
    if kind == "shufflenetv2":
        return boxes, ssdseglib.models.ShuffleNetV2SsdSegBuilder(
            input_image_shape=IMAGE_SHAPE, model_size='1x', use_additional_depthwise_convolution=True, use_residual_connections=True,
            number_of_boxes_per_point=[6, 6, 6, 6], number_of_classes=4,
            center_x_boxes_default=boxes.get_boxes_coordinates_center_x('ssd'), center_y_boxes_default=boxes.get_boxes_coordinates_center_y('ssd'),
            width_boxes_default=boxes.get_boxes_coordinates_width('ssd'), height_boxes_default=boxes.get_boxes_coordinates_height('ssd'),
            standard_deviations_centroids_offsets=STDS)
    builder = ssdseglib.models.MobileNetV2SsdSegBuilder(
        input_image_shape=IMAGE_SHAPE, number_of_boxes_per_point=[6, 6, 6, 6], number_of_classes=4,
        center_x_boxes_default=boxes.get_boxes_coordinates_center_x('ssd'), center_y_boxes_default=boxes.get_boxes_coordinates_center_y('ssd'),
        width_boxes_default=boxes.get_boxes_coordinates_width('ssd'), height_boxes_default=boxes.get_boxes_coordinates_height('ssd'),
        standard_deviations_centroids_offsets=STDS)
    return boxes, builder


def build_backbone_model():
    from ssdseglib import _graph as K
    _, b = build_models()
    inp = b._mobilenetv2_backbone()
    return K.Model(inputs=inp, outputs=[b._layers[n] for n in TAPS])


def build_full_model(kind="mobilenetv2"):
    import ssdseglib
    boxes, b = build_models(kind=kind)
    model = b.get_model_for_training('deeplabv3plus', 'ssdlite', segmentation_dilation_rates=(3, 6, 12))     # NB03#cell12
    model.compile(optimizer=ssdseglib.optimizers.Adam(learning_rate=1e-4),
                  loss={'output-mask': ssdseglib.losses.cross_entropy(classes_weights=CLASS_WEIGHTS),
                        'output-labels': ssdseglib.losses.confidence_loss, 'output-boxes': ssdseglib.losses.localization_loss},
                  loss_weights={'output-mask': 1.0, 'output-labels': 1.0, 'output-boxes': 1.0})                # NB03#cell14
    return boxes, model


def synthetic_images(batch, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (batch,) + IMAGE_SHAPE, dtype=np.uint8).astype(np.float32)


def synthetic_ground_truth(batch, seed, gmax=8):
    """1..8 boxes/image, labels 1..3, log-uniform sizes 24..400 px; masks rasterised from the boxes"""
    rng = np.random.default_rng(seed)
    h, w = IMAGE_SHAPE[:2]
    gt = np.zeros((batch, gmax, 5), np.float32)
    cnt = np.zeros(batch, np.int32)
    mask = np.zeros((batch, h, w), np.int64)
    for i in range(batch):
        g = int(rng.integers(1, gmax + 1))
        cnt[i] = g
        bw = np.minimum(np.exp(rng.uniform(np.log(24), np.log(400), g)), w - 2)
        bh = np.minimum(np.exp(rng.uniform(np.log(24), np.log(400), g)), h - 2)
        x0 = rng.uniform(0, w - 1 - bw)
        y0 = rng.uniform(0, h - 1 - bh)
        lab = rng.integers(1, 4, g)
        gt[i, :g] = np.stack([lab, x0, y0, x0 + bw, y0 + bh], axis=1)
        for l, xa, ya, ww, hh in zip(lab, x0, y0, bw, bh):
            mask[i, int(ya):int(ya + hh) + 1, int(xa):int(xa + ww) + 1] = l
    return gt, cnt, np.eye(4, dtype=np.float32)[mask]


class BackboneStep:
    """configs[1]: fwd + bwd (+ all-reduce) + Adam of the MobileNetV2 backbone on a resident batch"""
    workload = ("BASELINE.json configs[1]: MobileNetV2 backbone-only fwd+bwd(+all-reduce)+Adam, batch 32/GPU, 480x640x3, synthetic "
                "upstream gradients on the 3 head taps")

    def __init__(self, ctx, batch, rank, reducer):
        from ssdseglib import _engine as E
        self.model = build_backbone_model()
        self.eng = E.Engine(self.model, batch, training=True, ctx=ctx)
        self.reducer = reducer
        self.eng.set_input(synthetic_images(batch, 1993 + rank))
        rng = np.random.default_rng(7 + rank)
        self.seeds = [ctx.array(rng.standard_normal((batch,) + tuple(t.shape[1:]), dtype=np.float32) * np.float32(1e-3)) for t in self.model.outputs]

    def __call__(self):
        e = self.eng
        e.forward()
        for i, g in enumerate(self.seeds):
            e.seed_output_grad(i, g)
        e.backward_from_outputs()
        if self.reducer is not None:
            self.reducer()
        e.adam_step(lr=1e-4, grad_scale=self.reducer.scale if self.reducer is not None else 1.0)


class FullStep:
    """configs[2]: encode targets -> forward -> 3 losses -> backward (-> all-reduce) -> Adam"""
    workload = ("BASELINE.json configs[2]: full MobileNetV2-SSDLite-DeepLabV3+ train step (anchor encode + fwd + weighted CE / "
                "confidence+mining / localization losses + bwd (+all-reduce) + Adam), batch 32/GPU, 480x640x3, 9600 anchors, 4 classes")

    kind = "mobilenetv2"
    fix_q1 = False

    def __init__(self, ctx, batch, rank, reducer):
        from ssdseglib import _engine as E
        boxes, self.model = build_full_model(self.kind)
        if self.fix_q1:
            # SURVEY.md 8(d) config 5 allows either form: the reference's ShuffleNetV2 builder calls the head blocks without
            # relu_max_value, i.e. ReLU(max_value=0.0) (quirk Q1) -- every head activation and EVERY gradient of the model is then
            # exactly zero.  "Fixed": those layers clip at 6 like the MobileNetV2 variant's (what the author evidently meant).
            for l in self.model.layers:
                if type(l).__name__ == "ReLU" and l.max_value == 0.0:
                    l.max_value = 6.0
        self.eng = E.Engine(self.model, batch, training=True, ctx=ctx)
        self.eng.configure_losses(self.model._compiled["loss"], self.model._compiled["loss_weights"])
        self.reducer, self.ctx, self.batch = reducer, ctx, batch
        self.eng.set_input(synthetic_images(batch, 1993 + rank))
        gt, cnt, mask = synthetic_ground_truth(batch, 11 + rank)
        self.gt, self.cnt, self.gmax = ctx.array(gt), ctx.array(cnt), gt.shape[1]
        self.anchors = ctx.array(boxes.get_boxes_coordinates_corners('ssd'))
        self.det = self.eng.loss_ops["det"]
        self.eng.set_targets({'output-mask': mask, 'output-labels': self.det.y_labels, 'output-boxes': self.det.y_boxes})
        self.stds = (C.c_float * 4)(*STDS)

    def __call__(self):
        e = self.eng
        self.ctx.call("ssdseg_encode_targets", self.anchors, 9600, self.gt, self.cnt, self.batch, self.gmax, 4, 0.525, self.stds,
                      self.det.y_labels, self.det.y_boxes, None)
        e.forward()
        e.backward()
        if self.reducer is not None:
            self.reducer()
        e.adam_step(lr=1e-4, grad_scale=self.reducer.scale if self.reducer is not None else 1.0)


class ShuffleNetStep(FullStep):
    """configs[4] per GPU: the same train step on the ShuffleNetV2 1x variant (channel shuffle / split / max-pool kernels)"""
    workload = ("BASELINE.json configs[4] (per-GPU share): full ShuffleNetV2-1x-SSDLite-DeepLabV3+ train step (additional depthwise "
                "convolution + residual connections; anchor encode + fwd + 3 losses + bwd (+all-reduce) + Adam), batch 32/GPU, 480x640x3, "
                "9600 anchors, 4 classes; reference quirk Q1 (heads' ReLU max_value 0.0) kept bug-compatible")
    kind = "shufflenetv2"


class ShuffleNetFixedStep(ShuffleNetStep):
    """the same with quirk Q1 fixed (heads' ReLU6 instead of ReLU(max_value=0.0)): gradients are non-zero, kernel work identical"""
    workload = ShuffleNetStep.workload.replace("reference quirk Q1 (heads' ReLU max_value 0.0) kept bug-compatible",
                                               "reference quirk Q1 FIXED (heads' ReLU max_value 6.0 instead of 0.0), so that gradients are non-zero")
    fix_q1 = True


STEPS = {"backbone": BackboneStep, "full": FullStep, "shufflenet": ShuffleNetStep, "shufflenet-q1fixed": ShuffleNetFixedStep}


def cpu_baseline(workload, sample_batch=1):
    """NumPy oracle of the same step on the host cores, bounded sample (kind "port": TensorFlow cannot run here)."""
    from oracle import np_ops as O
    from oracle.np_model import NpModel
    x = synthetic_images(sample_batch, 1993)
    rng = np.random.default_rng(7)
    if workload == "backbone":
        model = build_backbone_model()
        ref = NpModel(model, dtype=np.float32)
        t0 = time.perf_counter()
        outs = ref.forward(x, training=True)
        gouts = [(rng.standard_normal(o.shape, dtype=np.float32) * np.float32(1e-3)) for o in outs]
    else:
        boxes, model = build_full_model("shufflenetv2" if workload.startswith("shufflenet") else "mobilenetv2")
        if workload == "shufflenet-q1fixed":
            for l in model.layers:
                if type(l).__name__ == "ReLU" and l.max_value == 0.0:
                    l.max_value = 6.0
        gt, cnt, mask = synthetic_ground_truth(sample_batch, 11)
        ref = NpModel(model, dtype=np.float32)
        corners = boxes.get_boxes_coordinates_corners('ssd')
        t0 = time.perf_counter()
        enc = [O.encode_targets(corners, gt[i, :cnt[i]], 4, 0.525, STDS) for i in range(sample_batch)]
        y_labels, y_boxes = np.stack([e[0] for e in enc]), np.stack([e[1] for e in enc])
        p_mask, p_labels, p_boxes = ref.forward(x, training=True)
        _, dmask = O.cross_entropy_loss(mask, p_mask, np.asarray(CLASS_WEIGHTS, np.float32))
        _, dconf, _ = O.confidence_loss(y_labels, p_labels)
        _, dloc = O.localization_loss(y_boxes, p_boxes)
        gouts = [dmask / sample_batch, dconf / sample_batch, dloc / sample_batch]
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
    return dict(value=round(sample_batch / dt, 4), unit="images/sec", cores=int(threads), kind="port",
                sample=f"{sample_batch} image(s) 480x640, the same {workload} step (fwd+bwd+Adam) through the NumPy oracle -- a CPU "
                       f"restatement of ssdseglib, not TensorFlow (not installable here); BLAS matmuls on {threads} threads, elementwise "
                       f"passes on 1; {dt:.1f} s")


def pmc_traffic(kernel: str, workload: str, batch: int):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same workload.

    FETCH_SIZE / WRITE_SIZE cannot be collected inside a timed run (separate `--pmc` passes, never together with other trace
    domains), so they come from profiles/<round>_pmc_traffic_<workload>_b<batch>.json -- written by scripts/gpu_measure.sh +
    scripts/pmc_summary.py with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md -- and are averaged over the
    launches of that kernel symbol exactly like `achieved`.  None when no summary for this workload/batch is committed or the
    kernel symbol is not in it."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", f"r*_pmc_traffic_{workload}_b{batch}.json")))
    if not files:
        return None, None
    try:
        kernels = json.load(open(files[-1]))["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None
    key = kernel.strip("()")
    row = kernels.get(key)
    if row is None and " [" in key:
        # the library labels some launches by role ("conv3_wino_kernel<true> [fwd]"); rocprofv3 knows the bare symbol
        row = kernels.get(key.split(" [")[0])
    if row is None:
        return None, None
    return round(row["hbm_bytes_per_launch"], 1), os.path.relpath(files[-1], here)


def achievable_ceiling(bound: str):
    """the committed micro-benchmark figure for this bound (scripts/micro/*.hip -> profiles/r*_microbench.txt), or None"""
    import glob
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    pat = "r*_mfma_f32_peak_microbench.txt" if bound == "mfma" else "r*_hbm_streaming_microbench.txt"
    files = sorted(glob.glob(os.path.join(here, "profiles", pat)))
    if not files:
        return None
    try:
        text = open(files[-1]).read()
    except OSError:
        return None
    vals = [float(v) for v in re.findall(r"([\d.]+) (?:TFLOP/s|TB/s)", text)]
    if not vals:
        return None
    if bound == "mfma":
        return {"value": max(vals), "unit": "TFLOP/s", "what": "register-only v_mfma_f32_32x32x2_f32 loop", "source": os.path.relpath(files[-1], here)}
    return {"value": round(max(vals) * 1e3, 1), "unit": "GB/s", "what": "best of the trivial float4 streams (copy, 3 reads : 1 write) over 1 GiB tensors",
            "source": os.path.relpath(files[-1], here)}


def nms_boxes_per_sec(ctx, batch, reps=20):
    """Secondary metric of BASELINE.json ("NMS boxes/sec"): B*9600 / t for box decode + combined NMS (4 classes
    incl. background, <= 4 per class, <= 10 per image) on synthetic head outputs drawn as 8(d) says -- probs = softmax(3 N(0,1)),
    offsets ~ U[0, 6) (the heads end in ReLU6, quirk Q3) -- at the reference's threshold pair (boxes_iou_threshold 0.025,
    labels_probability_threshold 0.725: NB03#cell23) = `value`, and at the stress pair 8(d) names (0.5, 0.05: thousands of
    candidates per class instead of tens)."""
    from ssdseglib import _hip as H
    boxes = default_boxes()
    cent = np.stack([boxes.get_boxes_coordinates_center_x('ssd'), boxes.get_boxes_coordinates_center_y('ssd'),
                     boxes.get_boxes_coordinates_width('ssd'), boxes.get_boxes_coordinates_height('ssd')], axis=1).astype(np.float32)
    a, c = cent.shape[0], 4
    rng = np.random.default_rng(1993)
    logits = (3.0 * rng.standard_normal((batch, a, c))).astype(np.float32)
    probs = np.exp(logits - logits.max(-1, keepdims=True))
    probs /= probs.sum(-1, keepdims=True)
    offs = rng.uniform(0, 6, (batch, a, 4)).astype(np.float32)
    d_off, d_cent, d_probs = ctx.array(offs), ctx.array(cent), ctx.array(probs)
    corners, out, valid = ctx.empty((batch, a, 4)), ctx.empty((batch, 10, 6)), ctx.empty(batch, np.int32)
    stds = (C.c_float * 4)(*STDS)

    def timed(iou_thr, score_thr):
        def run():
            ctx.call("ssdseg_decode_boxes", d_off, d_cent, batch, a, stds, corners)
            ctx.call("ssdseg_combined_nms", corners, d_probs, batch, a, c, 4, 10, iou_thr, score_thr, out, valid)
        run()
        ctx.sync()
        e0, e1 = H.Event(ctx), H.Event(ctx)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        ms = e0.elapsed_ms(e1) / reps
        return {"boxes_iou_threshold": iou_thr, "labels_probability_threshold": score_thr, "boxes_per_sec": round(batch * a / (ms * 1e-3), 1),
                "ms_per_batch": round(ms, 4), "candidates_per_image": round(float((probs > score_thr).sum()) / batch, 1),
                "detections_per_image": round(float(valid.download().mean()), 2)}

    ref, stress = timed(0.025, 0.725), timed(0.5, 0.05)
    return {"value": ref["boxes_per_sec"], "unit": "boxes/sec", "ms_per_batch": ref["ms_per_batch"],
            "what": f"decode + combined NMS of {batch} x {a} anchors x {c} classes (incl. background), HIP events over {reps} repetitions; "
                    f"value = the reference's thresholds (NB03#cell23)", "reference_thresholds": ref, "stress_thresholds": stress}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: start N FRESH rank processes of this same script (one per GPU;
    RANK / LOCAL_RANK / WORLD_SIZE and a private rendezvous path in their environment), relay rank 0's JSON line, return
    non-zero if any rank did.  The parent never initialises HIP (no re-exec of a process that touched the GPU), and the children
    are started, not exec'ed into.  When one rank dies the others would wait in the collective for ever: they are terminated (by
    pid) and the first failure's code is returned."""
    import shutil
    import subprocess
    import tempfile
    scratch = tempfile.mkdtemp(prefix="ssdseg_bench_")
    env = dict(os.environ, WORLD_SIZE=str(n), SSDSEG_RDZV_FILE=os.path.join(scratch, "rdzv"), SSDSEG_LAUNCHER="bench.py")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if env.get("SSDSEG_COMM") == "host":
        env.setdefault("SSDSEG_COMM_DIR", os.path.join(scratch, "comm"))
    procs = []
    out0_path = os.path.join(scratch, "rank0.stdout")
    try:
        with open(out0_path, "wb") as out0:
            for r in range(n):
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=out0 if r == 0 else None))
        rc, alive = 0, list(procs)
        while alive:
            time.sleep(0.2)
            for p in list(alive):
                if p.poll() is None:
                    continue
                alive.remove(p)
                if p.returncode != 0 and rc == 0:
                    rc = p.returncode
                    print(f"bench.py: rank {procs.index(p)} exited with {p.returncode}; stopping the other ranks", file=sys.stderr)
                    for q in alive:
                        q.terminate()
        with open(out0_path, "rb") as f:
            sys.stdout.write(f.read().decode(errors="replace"))
        sys.stdout.flush()
        return rc if rc >= 0 else 128 - rc
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(scratch, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--workload", default="full", choices=list(STEPS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--all-kernels", action="store_true", help="list every kernel symbol of the survey step (default: the 16 heaviest)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # nobody launched the ranks for us: this process becomes the launcher -- BEFORE anything here touches HIP
        sys.exit(launch_ranks(args.gpus))

    from ssdseglib import _hip as H
    from ssdseglib import _parallel as P
    rank, local_rank, world = P.env_world()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}", file=sys.stderr)
        sys.exit(2)

    # rehearsal switch (one-GPU box): SSDSEG_BENCH_DEVICE=0 puts every rank on one card -- then SSDSEG_COMM=host is needed too,
    # RCCL refuses two ranks on one device
    device = int(os.environ.get("SSDSEG_BENCH_DEVICE", local_rank))
    ctx = H.Context(device)
    comm = P.init_comm(ctx)                      # None for a single process; RCCL communicator otherwise

    step = STEPS[args.workload](ctx, args.batch, rank, None)
    if comm is not None:
        step.reducer = P.GradientAllReduce(comm, step.eng)
        comm.broadcast(step.eng.P["params"], 0)  # replicas start identical whatever the seeds did
        if step.eng.P["n_st"]:
            comm.broadcast(step.eng.P["state"], 0)

    def barrier():
        ctx.sync()
        if comm is not None:
            comm.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    survey, dominant = {}, None
    if not args.no_kernel_timing:
        # one untimed, fully instrumented step (every launch bracketed by HIP events) to rank the kernels ...
        # (the survey runs WITHOUT the side stream: with weight-gradient kernels co-running, per-kernel durations include each
        # other's interference and the ranking would pick whatever happened to be stretched most)
        ctx.side_enable(False)
        ctx.timing(True)
        ctx.timing_reset()
        step()
        survey = ctx.timing_report()
        ctx.side_enable(True)
        dominant = max(survey.items(), key=lambda kv: kv[1]["ms"])[0]
        # ... then, inside the timed region, only the dominant kernel symbol is bracketed (2 event records per launch of
        # that kernel; bracketing all ~600 launches/step would cost ~12 % of the step)
        ctx.timing_filter(dominant)
        ctx.timing_reset()
        barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    elapsed = time.perf_counter() - t0
    barrier()
    report = ctx.timing_report() if not args.no_kernel_timing else {}
    # The timed region runs weight-gradient kernels on a side stream, so the dominant kernel's HIP-event duration above includes
    # whatever co-ran with it.  A few extra steps WITHOUT the side stream give the same kernel's stand-alone duration; reported
    # as `roofline_isolated` next to (never instead of) the timed-region `roofline`.
    isolated = {}
    if report and dominant is not None:
        ctx.side_enable(False)
        step()
        ctx.sync()
        ctx.timing_reset()
        for _ in range(3):
            step()
        ctx.sync()
        isolated = ctx.timing_report()
        ctx.side_enable(True)
    ctx.timing(False)

    comm_world = 1
    if comm is not None:
        elapsed = comm.max(elapsed)              # the slowest rank defines the step
        comm_world = comm.group_size()           # (collective on the host transport: every rank calls it)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = args.batch * world * args.steps / elapsed
        out = {
            "metric": "images/sec (fwd+bwd, 480x640)", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": step.workload, "global_batch": args.batch * world, "per_gpu_batch": args.batch,
                       "parallelism": f"dp{world}", "collective": (comm.transport if comm is not None else None),
                       # the communicator's own idea of the group (ssdseg_comm_info for RCCL): proves the collective saw N ranks
                       "comm_world": comm_world, "launcher": os.environ.get("SSDSEG_LAUNCHER", "external"),
                       "device": ctx.device_name()},
        }
        if report:
            total_ms = sum(v["ms"] for v in survey.values())           # all kernels of the survey step
            name, dom = dominant, report[dominant]                       # dominant kernel: timed-region launches only

            def roof(d):
                """achieved = algorithmic bytes (or flops) per launch / average launch duration; the bound is whichever
                roof the kernel sits closer to"""
                avg_ms = d["ms"] / d["count"]
                gbs = d["bytes"] / d["count"] / (avg_ms * 1e-3) / 1e9
                tfs = d["flops"] / d["count"] / (avg_ms * 1e-3) / 1e12
                return avg_ms, gbs, tfs

            avg_ms, gbs, tfs = roof(dom)
            hbm_frac, mfma_frac = gbs / HBM_PEAK_GBS, tfs / MFMA_F32_PEAK_TFLOPS
            bound = "mfma" if mfma_frac > hbm_frac else "hbm"
            traffic, traffic_src = pmc_traffic(name, args.workload, args.batch)
            alg_bytes, view_bytes = dom["bytes"] / dom["count"], dom.get("view_bytes", 0.0) / dom["count"]
            out["roofline"] = {
                "kernel": name, "bound": bound,
                "achieved": round(tfs if bound == "mfma" else gbs, 2), "peak": MFMA_F32_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS,
                "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": round(mfma_frac if bound == "mfma" else hbm_frac, 4),
                "traffic": traffic, "traffic_source": traffic_src, "launches": dom["count"], "avg_launch_ms": round(avg_ms, 4),
                "share_of_kernel_time": round(survey[dominant]["ms"] / total_ms, 4),
                # depthwise bwd 4(2X+Y+18C), pointwise read X + write Y + W, dense 3x3 X + Y + W (never the im2col operand)
                "algorithmic_bytes_per_launch": alg_bytes, "flops_per_launch": dom["flops"] / dom["count"],
                # what this design reads ON TOP of 8(d)'s ideal: the raw forward output y next to g wherever dY is a BatchNorm-backward
                # gradient view (reported, never counted in `achieved`)
                "gradient_view_second_tensor_bytes_per_launch": view_bytes,
                "hbm_GBps_incl_view_bytes": round((alg_bytes + view_bytes) / (avg_ms * 1e-3) / 1e9, 1),
                "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic else None,
            }
            if name.startswith("conv3_wino"):
                # Winograd, fp32: F(2x2,3x3) executes 16 multiply-adds per 2x2 output tile and channel pair (F(4x4,3x3): 36 per 4x4
                # tile) where the direct form has 36 (144).  `achieved` / `frac` above count the flops EXECUTED on the MFMA pipe (a
                # utilisation, <= 1); the figure by the direct convolution's 18 m cin cout flops -- SURVEY.md 8(d)'s unit -- is here.
                f4 = name.startswith("conv3_wino4")
                ratio = 4.0 if f4 else 2.25
                out["roofline"]["algorithm"] = ("winograd F(4x4,3x3)" if f4 else "winograd F(2x2,3x3)") + ", fp32 transforms and accumulation"
                out["roofline"]["direct_equivalent"] = {"flops_per_launch": dom["flops"] / dom["count"] * ratio, "TFLOP/s": round(tfs * ratio, 2),
                                                        "over_peak": round(mfma_frac * ratio, 4)}
                # gfx950: the fp32 MFMA and the vector ALU are one pipe (profiles/r03_mfma_f32_filler_cost.txt) -- the transforms'
                # vector instructions are not hidden behind the MFMAs, they add to them; `frac` is MFMA time / kernel time
                out["roofline"]["note"] = "fp32 MFMA and VALU share one pipe on gfx950: 1 - frac includes the transform arithmetic"
            ach = achievable_ceiling(bound)
            if ach is not None:
                # extra context, not the contract's `peak`: what a trivial micro-benchmark sustains on this chip (committed summary)
                out["roofline"]["achievable_ceiling"] = ach
            if dominant in isolated and isolated[dominant]["count"] > 0:
                iso = isolated[dominant]
                iso_ms, igbs, itfs = roof(iso)
                out["roofline_isolated"] = {
                    "kernel": name, "note": "same kernel, 3 extra steps with the side stream disabled (no co-running weight-gradient kernel)",
                    "achieved": round(itfs if bound == "mfma" else igbs, 2), "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                    "frac": round((itfs / MFMA_F32_PEAK_TFLOPS) if bound == "mfma" else (igbs / HBM_PEAK_GBS), 4),
                    "launches": iso["count"], "avg_launch_ms": round(iso_ms, 4)}
            # the north-star's two roofline targets, each on the heaviest kernel of its class in the survey step (side stream off)
            def heaviest(pred):
                c = [(k, v) for k, v in survey.items() if pred(k) and v["ms"] > 0]
                return max(c, key=lambda kv: kv[1]["ms"]) if c else None
            dwk = heaviest(lambda k: k.lstrip("(").startswith("dw_bwd"))
            if dwk is not None:
                a, g, _ = roof(dwk[1])
                out["roofline_depthwise_bwd"] = {"kernel": dwk[0], "bound": "hbm", "achieved": round(g, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                 "frac": round(g / HBM_PEAK_GBS, 4), "launches": dwk[1]["count"], "avg_launch_ms": round(a, 4),
                                                 "note": "survey step (isolated); bytes = 4(2X+Y+18C) per SURVEY.md 8(d)"}
            top = sorted(survey.items(), key=lambda kv: -kv[1]["ms"])[:(None if args.all_kernels else 16)]
            out["kernels_survey_step"] = [
                {"kernel": k, "launches": v["count"], "ms_per_step": round(v["ms"], 4),
                 "GB/s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0.0,
                 "TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0.0} for k, v in top]
            out["kernel_ms_per_step"] = round(total_ms, 3)
        if world == 1 and args.workload != "backbone":
            out["nms_boxes_per_sec"] = nms_boxes_per_sec(ctx, args.batch)   # the metric's second figure (outside the timed region)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, 12 if args.workload == "backbone" else 6)   # ~10-30 s of host work (6.5 s per 3 images of the full step on the GPU box)
        print(json.dumps(out), flush=True)

    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
