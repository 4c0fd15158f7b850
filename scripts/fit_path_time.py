#!/usr/bin/env python3
"""PCIe-inclusive step time of the full train step as `fit` runs it: every batch handed over as host NumPy arrays
(images 118 MB, one-hot mask 157 MB, encoded labels/offsets 2 x 4.9 MB at batch 32), versus the resident-input step
bench.py times.  usage: python scripts/fit_path_time.py [batch]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))
import numpy as np
import bench
from ssdseglib import _engine as E, _hip as H

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = H.Context(0)
E.set_default_context(ctx)
boxes, model = bench.build_full_model()
x = bench.synthetic_images(batch, 1993)
gt, cnt, mask = bench.synthetic_ground_truth(batch, 11)
import ssdseglib
enc = ssdseglib.datacoder.DataEncoderDecoder(
    4, bench.IMAGE_SHAPE[:2], xmin_boxes_default=boxes.get_boxes_coordinates_xmin('ssd'), ymin_boxes_default=boxes.get_boxes_coordinates_ymin('ssd'),
    xmax_boxes_default=boxes.get_boxes_coordinates_xmax('ssd'), ymax_boxes_default=boxes.get_boxes_coordinates_ymax('ssd'),
    iou_threshold=0.525, standard_deviations_centroids_offsets=bench.STDS)
labels, offsets = enc.encode_batch([gt[i, :cnt[i]] for i in range(batch)])
y = {'output-mask': mask, 'output-labels': labels, 'output-boxes': offsets}
for _ in range(3):
    model.train_on_batch(x, y)
ctx.sync()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    logs = model.train_on_batch(x, y)      # uploads x and y, runs the step, downloads the three (B,) losses
dt = (time.perf_counter() - t0) / K
print(f"fit path (host arrays in, losses out): {dt * 1e3:.1f} ms/step = {batch / dt:.0f} images/sec; loss {logs['loss']:.4f}")
for mode in ("0", "1"):
    os.environ["SSDSEG_FIT_OVERLAP"] = mode
    model.fit([(x, y)] * 3, epochs=1)
    t0 = time.perf_counter()
    model.fit([(x, y)] * K, epochs=1)
    dtf = (time.perf_counter() - t0) / K
    print(f"fit(), SSDSEG_FIT_OVERLAP={mode} ({'next batch staged on the copy stream under the running step' if mode == '1' else 'synchronous hand-over'}): "
          f"{dtf * 1e3:.1f} ms/step = {batch / dtf:.0f} images/sec")
cb = ssdseglib.datacoder.CompactBatch(x.astype(np.uint8), mask.argmax(-1).astype(np.uint8), [gt[i, :cnt[i]] for i in range(batch)],
                                      (np.arange(batch) % 2).astype(np.uint8), enc)
for mode in ("0", "1"):
    os.environ["SSDSEG_FIT_OVERLAP"] = mode
    model.fit([cb] * 3, epochs=1)
    t0 = time.perf_counter()
    model.fit([cb] * K, epochs=1)
    dtc = (time.perf_counter() - t0) / K
    print(f"fit() on COMPACT batches ({(cb.images.nbytes + cb.mask_index.nbytes) / 1e6:.0f} MB uint8 up, expansion + flip + anchor encoding on the "
          f"device), SSDSEG_FIT_OVERLAP={mode}: {dtc * 1e3:.1f} ms/step = {batch / dtc:.0f} images/sec")
eng = E.engine_for(model, batch, True)
t0 = time.perf_counter()
for _ in range(K):
    eng.train_step(optimizer=model._compiled["optimizer"])
ctx.sync()
dt2 = (time.perf_counter() - t0) / K
print(f"resident inputs: {dt2 * 1e3:.1f} ms/step = {batch / dt2:.0f} images/sec; hand-over cost {1e3 * (dt - dt2):.1f} ms/step for "
      f"{(x.nbytes + mask.nbytes + labels.nbytes + offsets.nbytes) / 1e6:.0f} MB")
