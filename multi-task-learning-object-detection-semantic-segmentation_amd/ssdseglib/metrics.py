"""Training metrics with the reference's factory API (reference metrics.py:5-220): each factory returns
`metric(y_true, y_pred) -> (batch,)`, computed by the HIP kernels `ssdseg_metric_*`.

The returned callables carry `metric_kind` and their parameters, so `Model.compile(metrics={output: fn})` can evaluate them
inside the train / validation step straight from the device buffers of that step (the segmentation metric from the
low-resolution logits: `output-mask` is never materialised while training).  Called directly they accept host arrays
(uploaded) or device buffers.  Semantics follow the reference to the letter:
  * jaccard_iou_segmentation_masks_metric is the SOFT Jaccard on probabilities (no arg-max), epsilon in the denominator;
  * categorical_accuracy_metric counts equal(one_hot(argmax p), y_true) per class over ALL boxes (agreeing zeros count);
  * jaccard_iou_bounding_boxes_metric keeps the reference's decode conventions and is NaN for images without objects.
"""
import ctypes as C
from typing import Callable, List

import numpy as np


def _ctx():
    from . import _engine
    return _engine.default_context()


def _dev(ctx, a):
    from . import _hip as H
    return a if isinstance(a, H.DeviceBuffer) else ctx.array(np.ascontiguousarray(a, np.float32))


def jaccard_iou_segmentation_masks(classes_weights: List[float]) -> Callable:
    """reference metrics.py:5-50"""
    if len(classes_weights) != 4:
        raise ValueError("the segmentation metric kernel handles the reference's 4 classes (background + 3)")
    cw = (C.c_float * 4)(*[float(x) for x in classes_weights])

    def jaccard_iou_segmentation_masks_metric(y_true, y_pred):
        ctx = _ctx()
        y_pred = y_pred if hasattr(y_pred, "shape") else np.asarray(y_pred, np.float32)
        n, h, w, c = tuple(y_pred.shape)
        out = ctx.empty(n)
        ctx.call("ssdseg_metric_mask_iou", _dev(ctx, y_pred), n, h, w, c, 1, 1, 0, _dev(ctx, y_true), cw, out)
        return out.download()

    f = jaccard_iou_segmentation_masks_metric
    f.metric_kind, f.classes_weights, f._cw = "mask_iou", tuple(float(x) for x in classes_weights), cw
    return f


def jaccard_iou_bounding_boxes(center_x_boxes_default, center_y_boxes_default, width_boxes_default, height_boxes_default,
                               standard_deviations_centroids_offsets) -> Callable:
    """reference metrics.py:53-173"""
    anchors = np.stack([np.asarray(v, np.float32).reshape(-1) for v in
                        (center_x_boxes_default, center_y_boxes_default, width_boxes_default, height_boxes_default)], axis=1)
    stds = (C.c_float * 4)(*[float(s) for s in standard_deviations_centroids_offsets])
    state = {}

    def anchors_on(ctx):
        if state.get("ctx") is not ctx:
            state["ctx"], state["buf"] = ctx, ctx.array(np.ascontiguousarray(anchors))
        return state["buf"]

    def jaccard_iou_bounding_boxes_metric(y_true, y_pred):
        ctx = _ctx()
        y_pred = y_pred if hasattr(y_pred, "shape") else np.asarray(y_pred, np.float32)
        b, a = tuple(y_pred.shape)[:2]
        if a != anchors.shape[0]:
            raise ValueError(f"{a} boxes per image, {anchors.shape[0]} default boxes")
        out = ctx.empty(b)
        ctx.call("ssdseg_metric_box_iou", _dev(ctx, y_true), _dev(ctx, y_pred), anchors_on(ctx), stds, b, a, out)
        return out.download()

    f = jaccard_iou_bounding_boxes_metric
    f.metric_kind, f.anchors_on, f._stds = "box_iou", anchors_on, stds
    return f


def categorical_accuracy(classes_weights: List[float]) -> Callable:
    """reference metrics.py:176-220"""
    if len(classes_weights) != 4:
        raise ValueError("the label metric kernel handles the reference's 4 classes (background + 3)")
    cw = (C.c_float * 4)(*[float(x) for x in classes_weights])

    def categorical_accuracy_metric(y_true, y_pred):
        ctx = _ctx()
        y_pred = y_pred if hasattr(y_pred, "shape") else np.asarray(y_pred, np.float32)
        b, a, c = tuple(y_pred.shape)
        out = ctx.empty(b)
        ctx.call("ssdseg_metric_label_accuracy", _dev(ctx, y_true), _dev(ctx, y_pred), b, a, c, cw, out)
        return out.download()

    f = categorical_accuracy_metric
    f.metric_kind, f.classes_weights, f._cw = "label_accuracy", tuple(float(x) for x in classes_weights), cw
    return f
