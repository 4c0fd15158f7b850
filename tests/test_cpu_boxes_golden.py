"""Host-side anchor generator vs golden vectors captured by EXECUTING the reference's boxes.py
(scripts/make_golden_from_reference.py) -- bit-exact -- plus the reference's argument validation and quirk Q4."""
import json

import numpy as np
import pytest

GETTERS = ("corners", "xmin", "ymin", "xmax", "ymax", "centroids", "center_x", "center_y", "width", "height")


def _kwargs(meta):
    kw = {}
    for k, v in meta["ctor_kwargs"].items():
        kw[k] = tuple(tuple(x) if isinstance(x, list) else x for x in v) if isinstance(v, list) else v
    return kw


@pytest.mark.parametrize("name", ["nb03", "default", "ragged"])
def test_anchors_bit_exact(golden_dir, name):
    from ssdseglib.boxes import DefaultBoundingBoxes
    d = np.load(f"{golden_dir}/anchors_{name}.npz")
    meta = json.loads(str(d["meta"]))
    b = DefaultBoundingBoxes(**_kwargs(meta))
    assert np.array_equal(b.boxes_scales, d["boxes_scales"])
    b.rescale_boxes_coordinates(tuple(meta["image_shape"]))
    for g in GETTERS:
        got = getattr(b, f"get_boxes_coordinates_{g}")(coordinates_style="ssd")
        assert got.dtype == d[g].dtype and got.shape == d[g].shape
        assert np.array_equal(got, d[g]), f"{name}/{g}"
    for i, fm in enumerate(b.get_boxes_coordinates_corners("feature-maps")):
        assert np.array_equal(fm, d[f"fm_corners_{i}"])
    for i, fm in enumerate(b.get_boxes_coordinates_centroids("feature-maps")):
        assert np.array_equal(fm, d[f"fm_centroids_{i}"])


def test_nb03_anchor_count_and_order(golden_dir):
    d = np.load(f"{golden_dir}/anchors_nb03.npz")
    assert d["corners"].shape == (9600, 4)        # 7200 + 1800 + 480 + 120, six boxes per cell
    np.testing.assert_allclose(d["corners"][0], [-12.698076, -16.930172, 44.648075, 40.88017], rtol=1e-6)


def test_conversions_golden(golden_dir):
    from ssdseglib import boxes
    d = np.load(f"{golden_dir}/boxes_conversions.npz")
    cx, cy, w, h = boxes.coordinates_corners_to_centroids(d["xmin"], d["ymin"], d["xmax"], d["ymax"])
    for a, k in zip((cx, cy, w, h), ("cx", "cy", "w", "h")):
        assert np.array_equal(a, d[k])
    for a, k in zip(boxes.coordinates_centroids_to_corners(cx, cy, w, h), ("x0", "y0", "x1", "y1")):
        assert np.array_equal(a, d[k])


def test_argument_validation_like_reference():
    from ssdseglib.boxes import DefaultBoundingBoxes
    with pytest.raises(TypeError):     # an out-of-range FLOAT falls through to the tuple branch and fails iterating, as in the
        DefaultBoundingBoxes(((4, 4),), centers_padding_from_borders_percentage=0.5)          # reference (boxes.py:39-41)
    with pytest.raises(ValueError):
        DefaultBoundingBoxes(((4, 4),), centers_padding_from_borders_percentage=(0.5,))       # reference boxes.py:41-44
    with pytest.raises(ValueError):
        DefaultBoundingBoxes(((4, 4), (2, 2)), centers_padding_from_borders_percentage=(0.1,))
    with pytest.raises(ValueError):
        DefaultBoundingBoxes(((4, 4), (2, 2)), feature_maps_aspect_ratios=((1, 2),))           # reference boxes.py:54-55


def test_rescale_twice_compounds_like_reference():
    """quirk Q4 (reference boxes.py:162,171-177): the internal store is scaled in place"""
    from ssdseglib.boxes import DefaultBoundingBoxes
    b = DefaultBoundingBoxes(((4, 5),))
    b.rescale_boxes_coordinates((40, 50))
    once = b.get_boxes_coordinates_corners("ssd").copy()
    b.rescale_boxes_coordinates((40, 50))
    twice = b.get_boxes_coordinates_corners("ssd")
    fx, fy = np.float32(49 / 4), np.float32(39 / 3)
    assert np.allclose(twice[:, 0], once[:, 0] * fx) and np.allclose(twice[:, 1], once[:, 1] * fy)
