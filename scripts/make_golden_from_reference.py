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
    dump_model_summary()
