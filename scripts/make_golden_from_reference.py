#!/usr/bin/env python3
"""Generate golden fixtures from the *reference* repo (run in the build container only).

The reference (`/root/reference`) never travels to the GPU box, so everything the tests need from it is
captured here as data:

* anchors: outputs of executing the reference's NumPy-only `ssdseglib/boxes.py` (loaded by file path,
  because `ssdseglib/__init__.py` imports TensorFlow which is not installed) for
    - the NB03#cell6 configuration (the one trained/tested configuration), and
    - the NB01/default-argument configuration,
  stored as .npz (inputs = constructor kwargs, outputs = every getter's array).
* model summary: the `model.summary()` *output* stored inside NB03#cell12, parsed into JSON rows
  (layer name, type, output shape, #params, inbound layers) + the three totals.
* offline evaluators: outputs of executing the reference's `ssdseglib/evaluators.py` functions that are pure NumPy / csv --
  `_iou_boxes_pred_vs_true` (:6-62) and `average_precision_object_detection` (:65-186) -- on seeded synthetic predictions and
  ground-truth CSV files.  The module has `import tensorflow as tf` at its top (used only by the PNG reader of
  `jaccard_iou_semantic_segmentation`, which is NOT executed and stays "parity unpinned"); it is loaded by file path with an EMPTY
  placeholder module registered under that name for the duration of the load, so that the import statement succeeds.  Nothing
  of TensorFlow is emulated: the two functions run here never touch `tf`.

Usage: python scripts/make_golden_from_reference.py [/root/reference]
"""
import importlib.util
import json
import re
import sys
import warnings
from pathlib import Path

import numpy as np

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)


def load_reference_boxes():
    spec = importlib.util.spec_from_file_location("_ref_boxes", REF / "ssdseglib" / "boxes.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def dump_anchors(mod, name, ctor_kwargs, image_shape):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        b = mod.DefaultBoundingBoxes(**ctor_kwargs)
        b.rescale_boxes_coordinates(image_shape=image_shape)
        arrays = {}
        for g in ("corners", "xmin", "ymin", "xmax", "ymax", "centroids", "center_x", "center_y", "width", "height"):
            arrays[g] = getattr(b, f"get_boxes_coordinates_{g}")(coordinates_style="ssd")
        fm = b.get_boxes_coordinates_corners(coordinates_style="feature-maps")
        for i, a in enumerate(fm):
            arrays[f"fm_corners_{i}"] = a
        fmc = b.get_boxes_coordinates_centroids(coordinates_style="feature-maps")
        for i, a in enumerate(fmc):
            arrays[f"fm_centroids_{i}"] = a
        arrays["boxes_scales"] = np.asarray(b.boxes_scales)
    meta = dict(ctor_kwargs=ctor_kwargs, image_shape=list(image_shape))
    np.savez_compressed(OUT / f"anchors_{name}.npz", meta=json.dumps(meta), **arrays)
    print(name, {k: (v.shape, str(v.dtype)) for k, v in arrays.items() if not k.startswith("fm_")})


def dump_free_functions(mod):
    rng = np.random.default_rng(7)
    xmin = rng.uniform(-20, 600, 257).astype(np.float32)
    ymin = rng.uniform(-20, 440, 257).astype(np.float32)
    xmax = xmin + rng.uniform(1, 300, 257).astype(np.float32)
    ymax = ymin + rng.uniform(1, 300, 257).astype(np.float32)
    cx, cy, w, h = mod.coordinates_corners_to_centroids(xmin, ymin, xmax, ymax)
    x0, y0, x1, y1 = mod.coordinates_centroids_to_corners(cx, cy, w, h)
    np.savez_compressed(OUT / "boxes_conversions.npz", xmin=xmin, ymin=ymin, xmax=xmax, ymax=ymax,
                        cx=cx, cy=cy, w=w, h=h, x0=x0, y0=y0, x1=x1, y1=y1)


def load_reference_evaluators():
    import types
    saved = sys.modules.get("tensorflow")
    sys.modules["tensorflow"] = types.ModuleType("tensorflow")      # empty: satisfies the module-level import, nothing else
    try:
        spec = importlib.util.spec_from_file_location("_ref_evaluators", REF / "ssdseglib" / "evaluators.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if saved is None:
            del sys.modules["tensorflow"]
        else:
            sys.modules["tensorflow"] = saved
    return mod


def dump_evaluators(mod):
    """seeded detections of S samples (P boxes each after NMS, label 0 = background) against ground-truth CSV files
    (label,xmin,ymin,xmax,ymax per row; samples without objects: an empty file)"""
    import tempfile
    rng = np.random.default_rng(20240611)
    S, P, GMAX, classes, bg = 12, 10, 5, [0, 1, 2, 3], 0
    gt_cnt = rng.integers(0, GMAX + 1, S)
    gt_cnt[3] = 0                                                    # a sample without ground truth
    gt = np.zeros((S, GMAX, 5), np.float32)
    labels = np.zeros((S, P), np.int32)
    conf = np.round(rng.uniform(0.05, 1.0, (S, P)), 2).astype(np.float32)      # two decimals: ties in the ranking are likely
    boxes = np.zeros((S, P, 4), np.float32)
    for s in range(S):
        for g in range(gt_cnt[s]):
            x0, y0 = rng.uniform(0, 500), rng.uniform(0, 380)
            w, h = rng.uniform(20, 140), rng.uniform(20, 100)
            gt[s, g] = [rng.integers(1, 3), x0, y0, x0 + w, y0 + h]  # classes 1 and 2 only: class 3 never has ground truth
        for p in range(P):
            kind = rng.integers(0, 4)
            if kind == 0 or gt_cnt[s] == 0:                          # a free box with any label (incl. background)
                x0, y0 = rng.uniform(0, 500), rng.uniform(0, 380)
                boxes[s, p] = [x0, y0, x0 + rng.uniform(10, 140), y0 + rng.uniform(10, 100)]
                labels[s, p] = rng.integers(0, 4)
            else:                                                    # a jittered copy of a ground-truth box, right or wrong label
                g = rng.integers(0, gt_cnt[s])
                jit = rng.normal(0, 12.0 if kind == 1 else 3.0, 4)
                boxes[s, p] = gt[s, g, 1:] + jit
                labels[s, p] = gt[s, g, 0] if kind != 3 else 1 + (int(gt[s, g, 0]) % 3)
    out = dict(labels=labels, conf=conf, boxes=boxes, gt=gt, gt_cnt=gt_cnt.astype(np.int32), classes=np.asarray(classes), background=np.int32(bg))
    with tempfile.TemporaryDirectory() as d, warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)          # np.trapz under NumPy 2
        paths = []
        for s in range(S):
            path = Path(d) / f"gt_{s}.csv"
            with open(path, "w", newline="") as f:
                for g in range(gt_cnt[s]):
                    f.write(",".join([str(int(gt[s, g, 0]))] + [repr(float(v)) for v in gt[s, g, 1:]]) + "\n")
            paths.append(str(path))
        for thr in (0.5, 0.75, 0.3):
            ap = mod.average_precision_object_detection(labels, conf, boxes, thr, paths, classes, bg)
            assert sorted(ap) == [1, 2, 3]
            out[f"ap_{int(thr * 100)}"] = np.asarray([ap[c] for c in (1, 2, 3)], np.float64)
    # the IoU helper alone: one sample with ground truth, one without
    s = int(np.argmax(gt_cnt))
    out["iou_sample"] = np.int32(s)
    out["iou"] = mod._iou_boxes_pred_vs_true(labels[s], boxes[s], gt[s, :gt_cnt[s], 0].astype(np.int32), gt[s, :gt_cnt[s], 1:])
    out["iou_empty"] = mod._iou_boxes_pred_vs_true(labels[3], boxes[3], np.zeros((0,), np.int32), np.zeros((0,), np.float32))
    np.savez_compressed(OUT / "evaluators_ap.npz", **out)
    print("evaluators:", {k: out[k] for k in ("ap_50", "ap_75", "ap_30")}, "iou", out["iou"].shape, out["iou_empty"].shape)


def dump_model_summary():
    nb = json.load(open(REF / "03-multi-task-network-ssdlite-deeplabv3plus-training.ipynb"))
    text = "".join(nb["cells"][12]["outputs"][0]["text"])
    lines = text.split("\n")
    # Keras prints fixed-width columns; wrapped cells continue on following lines.
    header = next(i for i, l in enumerate(lines) if l.startswith(" Layer (type)"))
    c1 = lines[header].index("Output Shape")
    c2 = lines[header].index("Param #")
    c3 = lines[header].index("Connected to")
    rows, cur = [], None
    for l in lines[header + 2:]:
        if l.startswith("====") or l.startswith("____"):
            break
        if not l.strip():
            if cur:
                rows.append(cur)
                cur = None
            continue
        cells = [l[:c1], l[c1:c2], l[c2:c3], l[c3:]]
        if cur is None:
            cur = ["", "", "", ""]
        for k in range(4):
            cur[k] += cells[k].strip()
    if cur:
        rows.append(cur)
    layers = []
    for name_type, shape, params, conn in rows:
        m = re.match(r"^(.*?)\((\w+)\)$", name_type)
        name, ltype = m.group(1).strip(), m.group(2)
        shp = [None if s.strip() == "None" else int(s) for s in re.findall(r"None|\d+", shape)]
        inbound = re.findall(r"'([^'\[]+)\[", conn)
        layers.append(dict(name=name, type=ltype, output_shape=shp, params=int(params), inbound=inbound))
    totals = {k: int(re.search(rf"{k}: (\d+)", text).group(1)) for k in ("Total params", "Trainable params", "Non-trainable params")}
    json.dump(dict(source="NB03#cell12 output (model.summary())", layers=layers, totals=totals),
              open(OUT / "nb03_model_summary.json", "w"), indent=1)
    print("summary layers:", len(layers), totals, "sum params:", sum(l["params"] for l in layers))


if __name__ == "__main__":
    ref_boxes = load_reference_boxes()
    dump_anchors(ref_boxes, "nb03",
                 dict(feature_maps_shapes=((30, 40), (15, 20), (8, 10), (4, 5)),
                      centers_padding_from_borders_percentage=(0.025, 0.05, 0.075, 0.1),
                      boxes_scales=(0.15, 0.95), additional_square_box=True), (480, 640))
    dump_anchors(ref_boxes, "default",
                 dict(feature_maps_shapes=((24, 32), (12, 16), (6, 8), (3, 4), (1, 1))), (384, 512))
    dump_anchors(ref_boxes, "ragged",
                 dict(feature_maps_shapes=((7, 9), (3, 5), (1, 2)),
                      feature_maps_aspect_ratios=((1, 2), (1, 2, 3, 0.5), (1,)),
                      boxes_scales=(0.1, 0.8), centers_padding_from_borders_percentage=0.0,
                      additional_square_box=False), (113, 257))
    dump_free_functions(ref_boxes)
    dump_evaluators(load_reference_evaluators())
    dump_model_summary()
