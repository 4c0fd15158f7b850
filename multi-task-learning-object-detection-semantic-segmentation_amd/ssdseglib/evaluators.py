"""Offline evaluation of saved predictions against ground-truth files (reference evaluators.py:6-247).

Host-side NumPy / CSV / PNG post-processing in the reference too -- nothing here is on the GPU hot path.  Same function
names, arguments and return values:
  * `average_precision_object_detection`: per class, predictions of all samples ranked by confidence; a prediction is a true
    positive when its best IoU with a ground-truth box OF THE SAME LABEL reaches the threshold (several predictions may hit the
    same ground-truth box -- the reference does not mark boxes as used); AP = trapezoid area under precision over recall;
  * `jaccard_iou_semantic_segmentation`: the soft Jaccard of the predicted probabilities against one-hot PNG masks, averaged
    over the samples, background dropped from the result.
IoU conventions as everywhere in the reference: pixel-inclusive extents (+1), 1e-7 in the denominator.
"""
import csv
from typing import Dict, List

import numpy as np

_EPS = 1e-7


def _iou_boxes_pred_vs_true(labels_pred, boxes_pred, labels_true, boxes_true) -> np.ndarray:
    """(P, T) IoU of predicted vs ground-truth corner boxes, zeroed where the labels differ (reference :6-62); (P, 1) zeros when
    there is no ground truth"""
    boxes_pred = np.asarray(boxes_pred, np.float32).reshape(-1, 4)
    labels_pred = np.asarray(labels_pred).reshape(-1)
    labels_true = np.asarray(labels_true).reshape(-1)
    if labels_true.size == 0:
        return np.zeros((boxes_pred.shape[0], 1), np.float32)
    boxes_true = np.asarray(boxes_true, np.float32).reshape(-1, 4)
    p, t = boxes_pred[:, None, :], boxes_true[None, :, :]
    iw = np.maximum(0.0, np.minimum(p[..., 2], t[..., 2]) - np.maximum(p[..., 0], t[..., 0]) + 1.0)
    ih = np.maximum(0.0, np.minimum(p[..., 3], t[..., 3]) - np.maximum(p[..., 1], t[..., 1]) + 1.0)
    inter = iw * ih
    area_p = (p[..., 2] - p[..., 0] + 1.0) * (p[..., 3] - p[..., 1] + 1.0)
    area_t = (t[..., 2] - t[..., 0] + 1.0) * (t[..., 3] - t[..., 1] + 1.0)
    iou = inter / (area_p + area_t - inter + np.float32(_EPS))
    return (iou * (labels_pred[:, None] == labels_true[None, :])).astype(np.float32)


def _read_labels_boxes(path: str):
    labels, boxes = [], []
    with open(path, "r", newline="") as f:
        for row in csv.reader(f):
            if not row:
                continue
            labels.append(int(row[0]))
            boxes.append([float(v) for v in row[1:5]])
    return np.asarray(labels, np.int32), np.asarray(boxes, np.float32).reshape(-1, 4)


def average_precision_object_detection(labels_pred_batch, confidences_pred_batch, boxes_pred_batch, iou_threshold: float,
                                       path_files_labels_boxes: List[str], labels_codes: List[int], label_code_background: int) -> Dict[int, float]:
    """reference evaluators.py:65-186"""
    classes = [l for l in labels_codes if l != label_code_background]
    hits = {l: [] for l in classes}       # per class: (is true positive, confidence) of every prediction of that class
    n_true = {l: 0 for l in classes}
    for path, lab, conf, box in zip(path_files_labels_boxes, labels_pred_batch, confidences_pred_batch, boxes_pred_batch):
        lt, bt = _read_labels_boxes(path)
        for l in lt:
            n_true[int(l)] += 1
        lab = np.asarray(lab).reshape(-1)
        conf = np.asarray(conf, np.float32).reshape(-1)
        box = np.asarray(box, np.float32).reshape(-1, 4)
        keep = lab != label_code_background
        lab, conf, box = lab[keep], conf[keep], box[keep]
        if lab.size == 0:
            continue
        best = _iou_boxes_pred_vs_true(lab, box, lt, bt).max(axis=1)
        for l, c, tp in zip(lab, conf, best >= iou_threshold):
            hits[int(l)].append((float(tp), float(c)))
    out = {}
    for l in classes:
        if n_true[l] == 0 or not hits[l]:
            out[l] = 0.0
            continue
        h = np.asarray(hits[l], np.float32)
        order = np.argsort(h[:, 1])[::-1]                 # descending confidence, the reference's tie order
        tp = np.cumsum(h[order, 0])
        precision = tp / np.arange(1, tp.size + 1)
        recall = tp / n_true[l]
        out[l] = float(np.sum((recall[1:] - recall[:-1]) * (precision[1:] + precision[:-1]) * 0.5))   # np.trapz(y=precision, x=recall)
    return out


def jaccard_iou_semantic_segmentation(masks_pred_batch, path_files_masks: List[str], labels_codes: List[int],
                                      label_code_background: int) -> Dict[int, float]:
    """reference evaluators.py:189-247"""
    from PIL import Image
    pred = np.asarray(masks_pred_batch, np.float32)
    n_cls = len(labels_codes)
    classes = np.arange(n_cls)
    # one-hot of the single-channel PNG (pixel value = class label; values >= n_cls give an all-zero pixel, as tf.one_hot does)
    true = np.stack([(np.asarray(Image.open(p).convert("L"), np.int64)[..., None] == classes).astype(np.float32) for p in path_files_masks])
    inter = (true * pred).sum(axis=(1, 2))
    total = (true + pred).sum(axis=(1, 2))
    iou = (inter / (total - inter + np.float32(_EPS))).mean(axis=0)
    return {l: float(v) for l, v in zip(labels_codes, iou) if l != label_code_background}
