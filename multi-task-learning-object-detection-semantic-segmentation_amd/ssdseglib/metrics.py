"""Training metrics with the reference's factory API (reference metrics.py:5-220): each factory returns
`metric(y_true, y_pred) -> (batch,)`.  They are tiny host-side reductions for `History` (outside the
roofline-judged hot path), written in NumPy on downloaded outputs.
"""
from typing import Callable, List

import numpy as np


def jaccard_iou_segmentation_masks(classes_weights: List[float]) -> Callable:
    """weighted IoU of the arg-max masks (reference metrics.py:5-50)."""
    w = np.asarray(classes_weights, np.float32)[None, :]

    def jaccard_iou_segmentation_masks_metric(y_true, y_pred):
        y_true, y_pred = np.asarray(y_true, np.float32), np.asarray(y_pred, np.float32)
        c = y_pred.shape[-1]
        pred = np.eye(c, dtype=np.float32)[y_pred.argmax(-1)]
        inter = (y_true * pred).sum(axis=(1, 2))
        union = (y_true + pred).sum(axis=(1, 2)) - inter
        with np.errstate(invalid="ignore", divide="ignore"):
            iou = np.where(union > 0, inter / np.where(union > 0, union, 1), 1.0)
        return (iou * w).sum(-1)

    return jaccard_iou_segmentation_masks_metric


def jaccard_iou_bounding_boxes(center_x_boxes_default, center_y_boxes_default, width_boxes_default, height_boxes_default,
                               standard_deviations_centroids_offsets) -> Callable:
    """mean IoU between decoded predicted and ground-truth boxes over non-background anchors (reference metrics.py:53-173;
    like the reference it is NaN for images without objects, quirk Q10)."""
    cx, cy = np.asarray(center_x_boxes_default, np.float32), np.asarray(center_y_boxes_default, np.float32)
    aw, ah = np.asarray(width_boxes_default, np.float32), np.asarray(height_boxes_default, np.float32)
    sx, sy, sw, sh = (np.float32(s) for s in standard_deviations_centroids_offsets)

    def decode(o):
        x = o[..., 0] * sx * aw + cx
        y = o[..., 1] * sy * ah + cy
        w = (np.exp(o[..., 2] * sw) - 1.0) * aw
        h = (np.exp(o[..., 3] * sh) - 1.0) * ah
        return x - (w - 1) / 2, y - (h - 1) / 2, x + (w - 1) / 2, y + (h - 1) / 2

    def jaccard_iou_bounding_boxes_metric(y_true, y_pred):
        y_true, y_pred = np.asarray(y_true, np.float32), np.asarray(y_pred, np.float32)
        nb = (np.abs(y_true).sum(-1) > 0).astype(np.float32)
        tx0, ty0, tx1, ty1 = decode(y_true)
        px0, py0, px1, py1 = decode(y_pred)
        iw = np.maximum(0.0, np.minimum(tx1, px1) - np.maximum(tx0, px0) + 1.0)
        ih = np.maximum(0.0, np.minimum(ty1, py1) - np.maximum(ty0, py0) + 1.0)
        inter = iw * ih
        union = (tx1 - tx0 + 1.0) * (ty1 - ty0 + 1.0) + (px1 - px0 + 1.0) * (py1 - py0 + 1.0) - inter
        with np.errstate(invalid="ignore", divide="ignore"):
            return (inter / union * nb).sum(-1) / nb.sum(-1)

    return jaccard_iou_bounding_boxes_metric


def categorical_accuracy(classes_weights: List[float]) -> Callable:
    """class-weighted accuracy of the arg-max labels (reference metrics.py:176-220)."""
    w = np.asarray(classes_weights, np.float32)

    def categorical_accuracy_metric(y_true, y_pred):
        y_true, y_pred = np.asarray(y_true, np.float32), np.asarray(y_pred, np.float32)
        t, p = y_true.argmax(-1), y_pred.argmax(-1)
        c = y_true.shape[-1]
        out = np.zeros(y_true.shape[0], np.float32)
        for k in range(c):
            sel = t == k
            n = sel.sum(-1)
            with np.errstate(invalid="ignore", divide="ignore"):
                acc = np.where(n > 0, ((p == k) & sel).sum(-1) / np.where(n > 0, n, 1), 0.0)
            out += w[k] * acc
        return out

    return categorical_accuracy_metric
