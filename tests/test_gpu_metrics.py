"""GPU parity of the training metrics (reference metrics.py:5-220): the three kernels against the NumPy
restatement, and their wiring into compile(metrics=...) / train_on_batch / fit with Keras' history keys (NB03#cell14,16)."""
import ctypes as C

import numpy as np
import pytest

from oracle import np_ops as O
from oracle.np_model import NpModel
from tests.test_gpu_full_model import CW, STDS, build, make_targets

pytestmark = pytest.mark.gpu

LW = (0.0, 1 / 3, 1 / 3, 1 / 3)    # NB03#cell14


@pytest.mark.parametrize("n,h,w", [(2, 12, 16), (3, 60, 80), (1, 1, 1)])
def test_mask_iou_from_probabilities(ctx, rng, n, h, w):
    import ssdseglib
    from ssdseglib import _engine as E
    E.set_default_context(ctx)
    p = O.softmax(rng.normal(0, 2, (n, h, w, 4))).astype(np.float32)
    t = np.eye(4, dtype=np.float32)[rng.integers(0, 4, (n, h, w))]
    t[0, 0, 0] = 0                                       # a pixel without any class: formula is on values, not on an arg-max
    m = ssdseglib.metrics.jaccard_iou_segmentation_masks(CW)
    assert np.abs(m(t, p) - O.metric_mask_iou(t, p, CW)).max() < 1e-5
    assert np.abs(m(t, t) - O.metric_mask_iou(t, t, CW)).max() < 1e-6     # perfect prediction: ~ sum of the weights of present classes


@pytest.mark.parametrize("n,h,w,f", [(2, 6, 8, 4), (1, 30, 40, 4), (2, 5, 3, 2)])
def test_mask_iou_from_logits(ctx, rng, n, h, w, f):
    """the training path: soft Jaccard against softmax(bilinear x f (logits)) recomputed per pixel, nothing stored"""
    logits = rng.normal(0, 2, (n, h, w, 4)).astype(np.float32)
    t = np.eye(4, dtype=np.float32)[rng.integers(0, 4, (n, h * f, w * f))]
    p_ref = O.softmax(O.bilinear_fwd(logits.astype(np.float64), f, f))
    out = ctx.empty(n)
    ctx.call("ssdseg_metric_mask_iou", ctx.array(logits), n, h, w, 4, f, f, 1, ctx.array(t), (C.c_float * 4)(*CW), out)
    assert np.abs(out.download() - O.metric_mask_iou(t, p_ref, CW)).max() < 1e-5


@pytest.mark.parametrize("b,a", [(3, 500), (2, 9600), (1, 7)])
def test_label_accuracy(ctx, rng, b, a):
    import ssdseglib
    from ssdseglib import _engine as E
    E.set_default_context(ctx)
    p = O.softmax(rng.normal(0, 1, (b, a, 4))).astype(np.float32)
    p[0, :3] = 0.25                                      # exact ties: arg-max takes the first class
    t = np.eye(4, dtype=np.float32)[rng.integers(0, 4, (b, a))]
    m = ssdseglib.metrics.categorical_accuracy(LW)
    assert np.abs(m(t, p) - O.metric_label_accuracy(t, p, LW)).max() < 1e-6


@pytest.mark.parametrize("b,a,pos", [(3, 600, 0.05), (2, 9600, 0.01), (2, 50, 0.0)])
def test_box_iou(ctx, rng, b, a, pos):
    import ssdseglib
    from ssdseglib import _engine as E
    E.set_default_context(ctx)
    cx, cy = rng.uniform(0, 640, a).astype(np.float32), rng.uniform(0, 480, a).astype(np.float32)
    aw, ah = rng.uniform(20, 300, a).astype(np.float32), rng.uniform(20, 300, a).astype(np.float32)
    is_pos = rng.uniform(size=(b, a)) < pos
    t = (rng.normal(0, 2, (b, a, 4)) * is_pos[..., None]).astype(np.float32)
    p = rng.uniform(0, 6, (b, a, 4)).astype(np.float32)              # head outputs pass ReLU6 (quirk Q3)
    p[0, : a // 2] = t[0, : a // 2]                                  # some perfect predictions
    m = ssdseglib.metrics.jaccard_iou_bounding_boxes(cx, cy, aw, ah, STDS)
    got, ref = m(t, p), O.metric_box_iou(t, p, cx, cy, aw, ah, STDS)
    assert np.array_equal(np.isnan(got), np.isnan(ref))             # images without objects: NaN like the reference
    ok = ~np.isnan(ref)
    assert np.abs(got[ok] - ref[ok]).max() < 1e-5 if ok.any() else True
    with pytest.raises(ValueError):
        m(t[:, :-1], p[:, :-1])


def test_metrics_inside_train_step_and_fit(ctx, rng):
    """compile(metrics=...) as NB03#cell14: history keys and values == the oracle's metrics of the same forward pass"""
    import ssdseglib
    from ssdseglib import _engine as E
    E.set_default_context(ctx)
    batch = 3
    boxes, builder, model = build()
    enc, gts, targets = make_targets(rng, boxes, batch)
    x = rng.integers(0, 256, (batch, 96, 128, 3)).astype(np.float32)
    cxywh = [boxes.get_boxes_coordinates_center_x('ssd'), boxes.get_boxes_coordinates_center_y('ssd'),
             boxes.get_boxes_coordinates_width('ssd'), boxes.get_boxes_coordinates_height('ssd')]
    model.compile(optimizer=ssdseglib.optimizers.Adam(learning_rate=1e-4),
                  loss={'output-mask': ssdseglib.losses.cross_entropy(classes_weights=CW), 'output-labels': ssdseglib.losses.confidence_loss,
                        'output-boxes': ssdseglib.losses.localization_loss},
                  loss_weights={'output-mask': 1.0, 'output-labels': 1.0, 'output-boxes': 1.0},
                  metrics={'output-mask': ssdseglib.metrics.jaccard_iou_segmentation_masks(classes_weights=CW),
                           'output-labels': ssdseglib.metrics.categorical_accuracy(classes_weights=LW),
                           'output-boxes': ssdseglib.metrics.jaccard_iou_bounding_boxes(*cxywh, STDS)})
    ref = NpModel(model, dtype=np.float64)                            # the weights BEFORE the step
    p_mask, p_labels, p_boxes = ref.forward(x, training=True)
    logs = model.train_on_batch(x, targets)
    want = {'output-mask_jaccard_iou_segmentation_masks_metric': O.metric_mask_iou(targets['output-mask'], p_mask, CW).mean(),
            'output-labels_categorical_accuracy_metric': O.metric_label_accuracy(targets['output-labels'], p_labels, LW).mean(),
            'output-boxes_jaccard_iou_bounding_boxes_metric': O.metric_box_iou(targets['output-boxes'], p_boxes, *cxywh, STDS).mean()}
    for k, v in want.items():
        assert k in logs, (k, sorted(logs))
        assert (np.isnan(v) and np.isnan(logs[k])) or abs(logs[k] - v) < 2e-4 * max(1.0, abs(v)), (k, logs[k], v)
    hist = model.fit([(x, targets)], epochs=2, validation_data=[(x, targets)], verbose=0)
    for k in want:
        assert len(hist.history[k]) == 2 and len(hist.history["val_" + k]) == 2
    assert {"loss", "val_loss", "output-mask_loss", "val_output-boxes_loss"} <= set(hist.history)
