"""C-ABI shape checks without a GPU: the library loads, exports every symbol include/ssdseg.h declares, and the
host-side API mirrors the reference's error behaviour.  No compute call is made."""
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from ssdseglib import _hip
    if not os.path.exists(_hip.library_path()):
        g.build()
    return _hip.load_library()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(REPO, "include", "ssdseg.h")).read()
    declared = set(re.findall(r"\b(ssdseg_[a-z0-9_]+)\s*\(", header)) - {"ssdseg_ctx", "ssdseg_view", "ssdseg_gview"}
    assert len(declared) > 50
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in ssdseg.h but not exported: {missing}"
    from ssdseglib import _hip
    unbound = sorted(declared - set(_hip.declared_symbols()))
    assert not unbound, f"declared in ssdseg.h but without a ctypes signature: {unbound}"
    assert lib.ssdseg_version() == 1


def test_argument_errors_do_not_need_a_device(lib):
    import ctypes as C
    out = C.c_int()
    assert lib.ssdseg_dwconv_parts(1, 8, 8, 6, 1, 1, C.byref(out)) == -1000 - 4       # channels must be a multiple of 4
    assert b"invalid argument 4" in lib.ssdseg_last_error()
    assert lib.ssdseg_pwconv_parts(640, 96, C.byref(out)) == 0 and out.value == 5
    assert lib.ssdseg_stem_conv_parts(2, 480, 640, 32, C.byref(out)) == 0 and out.value > 0


def test_no_gpu_means_loud_failure_not_fallback():
    """the product path must not degrade to a CPU implementation"""
    from ssdseglib import _hip
    import ctypes as C
    n = C.c_int()
    rc = _hip.load_library().ssdseg_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_hip.SsdsegError):
        _hip.Context(0)
    import ssdseglib
    with pytest.raises(_hip.SsdsegError):
        ssdseglib.losses.localization_loss(np.zeros((1, 4, 4), np.float32), np.zeros((1, 4, 4), np.float32))


def test_datacoder_constructor_like_reference(golden_dir):
    import ssdseglib
    d = np.load(f"{golden_dir}/anchors_nb03.npz")
    corners = dict(xmin_boxes_default=d["xmin"], ymin_boxes_default=d["ymin"], xmax_boxes_default=d["xmax"], ymax_boxes_default=d["ymax"])
    cents = dict(center_x_boxes_default=d["center_x"], center_y_boxes_default=d["center_y"], width_boxes_default=d["width"],
                 height_boxes_default=d["height"])
    enc = ssdseglib.datacoder.DataEncoderDecoder(4, (480, 640), iou_threshold=0.525, **corners)
    assert np.array_equal(enc.center_x_boxes_default, d["center_x"]) and np.array_equal(enc.width_boxes_default, d["width"])
    enc2 = ssdseglib.datacoder.DataEncoderDecoder(4, (480, 640), **cents)
    assert np.allclose(enc2.xmin_boxes_default, d["xmin"], atol=1e-3)
    with pytest.raises(ValueError):
        ssdseglib.datacoder.DataEncoderDecoder(4, (480, 640), xmin_boxes_default=d["xmin"])          # reference datacoder.py:55-56
    with pytest.raises(ValueError):
        ssdseglib.datacoder.DataEncoderDecoder(4, (480, 640), center_x_boxes_default=d["center_x"])  # :74-75
    with pytest.raises(ValueError):
        ssdseglib.datacoder.DataEncoderDecoder(4, (480, 640), **corners, **cents)                    # quirk Q5 (:92-108)
    # ground-truth decode helpers (host side) agree with the oracle
    from oracle import np_ops as O
    off = np.zeros((9600, 4), np.float32)
    off[5] = [0.3, -0.2, 1.0, 2.0]
    got = enc.decode_to_centroids(off)
    assert np.allclose(got, O.decode_to_centroids_gt(off, d["centroids"], (0.1, 0.1, 0.2, 0.2)), atol=1e-4)
    assert np.all(enc.decode_to_corners(off)[0] == 0) and np.any(enc.decode_to_corners(off)[5] != 0)
    flipped = enc._flip_boxes(np.array([[1, 10, 20, 110, 220]], np.float32))
    assert flipped.tolist() == [[1, 530, 20, 630, 220]]                                              # x -> W - x (quirk Q8)


def test_metrics_and_loss_factories_shapes():
    import ssdseglib
    # the metric factories only describe the metric here: evaluating one needs the HIP library (tests/test_gpu_metrics.py)
    m = ssdseglib.metrics.jaccard_iou_segmentation_masks((0.05, 0.575, 0.135, 0.24))
    assert m.metric_kind == "mask_iou" and m.__name__ == "jaccard_iou_segmentation_masks_metric"
    acc = ssdseglib.metrics.categorical_accuracy((0.0, 1 / 3, 1 / 3, 1 / 3))
    assert acc.metric_kind == "label_accuracy" and acc.__name__ == "categorical_accuracy_metric"
    z = np.zeros(6, np.float32)
    iou = ssdseglib.metrics.jaccard_iou_bounding_boxes(z, z, z + 1, z + 1, (0.1, 0.1, 0.2, 0.2))
    assert iou.metric_kind == "box_iou" and iou.__name__ == "jaccard_iou_bounding_boxes_metric"
    with pytest.raises(ValueError):
        ssdseglib.metrics.categorical_accuracy((0.5, 0.5))
    fn = ssdseglib.losses.cross_entropy((0.05, 0.575, 0.135, 0.24))
    assert fn.loss_kind == "cross_entropy" and fn.classes_weights == (0.05, 0.575, 0.135, 0.24)
    opt = ssdseglib.optimizers.Adam(learning_rate=1e-4)
    assert (opt.learning_rate, opt.beta_1, opt.beta_2, opt.epsilon) == (1e-4, 0.9, 0.999, 1e-7)


def test_offline_evaluators_hand_worked(tmp_path):
    """ssdseglib.evaluators (reference evaluators.py:65-247) on a case small enough to do by hand"""
    import ssdseglib
    from PIL import Image
    # two samples; class 1 has 2 ground-truth boxes in total, class 2 has 1
    f0, f1 = tmp_path / "a.csv", tmp_path / "b.csv"
    f0.write_text("1,10,10,29,29\r\n2,50,50,69,69\r\n")
    f1.write_text("1,0,0,19,19\r\n")
    labels = np.array([[1, 1, 0], [1, 2, 0]])                       # 0 = background: ignored
    conf = np.array([[0.9, 0.6, 0.99], [0.8, 0.7, 0.5]], np.float32)
    boxes = np.array([[[10, 10, 29, 29], [100, 100, 119, 119], [0, 0, 5, 5]],          # exact hit, miss, (background)
                      [[0, 0, 19, 19], [0, 0, 19, 19], [0, 0, 1, 1]]], np.float32)   # exact hit, wrong label for that box
    ap = ssdseglib.evaluators.average_precision_object_detection(labels, conf, boxes, 0.5, [str(f0), str(f1)], [0, 1, 2], 0)
    # class 1 ranked by confidence: 0.9 TP, 0.8 TP, 0.6 FP -> precision (1, 1, 2/3), recall (0.5, 1, 1) -> area 0.5 * 1 = 0.5
    assert abs(ap[1] - 0.5) < 1e-6 and ap[2] == 0.0 and 0 not in ap
    iou = ssdseglib.evaluators._iou_boxes_pred_vs_true([1], [[0, 0, 9, 9]], [1, 2], [[5, 0, 14, 9], [0, 0, 9, 9]])
    assert np.allclose(iou, [[50 / 150, 0.0]], atol=1e-6)           # 5x10 overlap of two 10x10 boxes; label mismatch -> 0
    # segmentation: 2x2 image, mask classes [[0, 1], [1, 2]], prediction = the one-hot truth except one pixel split 50/50
    m = tmp_path / "m.png"
    Image.fromarray(np.array([[0, 1], [1, 2]], np.uint8)).save(m)
    pred = np.eye(3, dtype=np.float32)[np.array([[0, 1], [1, 2]])][None]
    pred[0, 0, 1] = [0.5, 0.5, 0.0]
    got = ssdseglib.evaluators.jaccard_iou_semantic_segmentation(pred, [str(m)], [0, 1, 2], 0)
    assert abs(got[1] - 1.5 / 2.0) < 1e-6 and abs(got[2] - 1.0) < 1e-6 and 0 not in got   # class 1: inter 1.5, total 3.5


def test_offline_evaluators_match_the_reference_fixture(tmp_path):
    """ssdseglib.evaluators against outputs of the REFERENCE's own `_iou_boxes_pred_vs_true` / `average_precision_object_detection`
    (evaluators.py:6-186, pure NumPy + csv) executed on seeded detections by scripts/make_golden_from_reference.py ->
    tests/golden/evaluators_ap.npz: 12 samples x 10 boxes after NMS (label 0 = background), 0-5 ground-truth boxes each (one
    sample with an empty file, class 3 without any ground truth, confidences with ties, several predictions on one ground-truth
    box -- the reference does not mark boxes as used, so recall and AP can exceed 1: 1.0088 at IoU 0.3).  The segmentation
    Jaccard reads its masks through TensorFlow in the reference and was not executed: that one stays hand-worked (above)."""
    import ssdseglib
    g = np.load(os.path.join(REPO, "tests", "golden", "evaluators_ap.npz"))
    labels, conf, boxes, gt, cnt = g["labels"], g["conf"], g["boxes"], g["gt"], g["gt_cnt"]
    paths = []
    for s in range(labels.shape[0]):
        path = tmp_path / f"gt_{s}.csv"
        with open(path, "w", newline="") as f:
            for k in range(cnt[s]):
                f.write(",".join([str(int(gt[s, k, 0]))] + [repr(float(v)) for v in gt[s, k, 1:]]) + "\n")
        paths.append(str(path))
    classes, bg = [int(c) for c in g["classes"]], int(g["background"])
    for thr in (0.5, 0.75, 0.3):
        ap = ssdseglib.evaluators.average_precision_object_detection(labels, conf, boxes, thr, paths, classes, bg)
        want = g[f"ap_{int(thr * 100)}"]
        assert sorted(ap) == [1, 2, 3]
        assert np.allclose([ap[1], ap[2], ap[3]], want, rtol=0, atol=1e-6), (thr, ap, want)
    assert g["ap_30"][0] > 1.0                       # the quirk is in the fixture, and reproduced
    s = int(g["iou_sample"])
    iou = ssdseglib.evaluators._iou_boxes_pred_vs_true(labels[s], boxes[s], gt[s, :cnt[s], 0].astype(np.int32), gt[s, :cnt[s], 1:])
    assert iou.shape == g["iou"].shape and np.allclose(iou, g["iou"], rtol=0, atol=1e-6)
    empty = ssdseglib.evaluators._iou_boxes_pred_vs_true(labels[3], boxes[3], np.zeros((0,), np.int32), np.zeros((0,), np.float32))
    assert empty.shape == g["iou_empty"].shape and not empty.any()


def _header_prototypes():
    """{function name: [C parameter type, ...]} parsed from include/ssdseg.h"""
    header = open(os.path.join(REPO, "include", "ssdseg.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(ssdseg_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", header, flags=re.S):
        params = [p.strip() for p in m.group(2).replace("\n", " ").split(",")]
        params = [] if params == ["void"] else params
        protos[m.group(1)] = [re.sub(r"\s+[A-Za-z_][A-Za-z0-9_]*$", "", p).replace(" *", "*").strip() for p in params]
    return protos


def test_ctypes_signatures_match_the_header(lib):
    """every parameter of every entry point has the ctypes type its C type demands -- in particular every `ssdseg_view*` /
    `ssdseg_gview*` is POINTER(struct), so handing a bare device pointer where a view struct is expected fails in Python
    (ctypes.ArgumentError) instead of making the library read device memory as a host struct (round 1: a GPU abort at the next
    synchronisation, DESIGN.md 'Fault log')."""
    import ctypes as C
    from ssdseglib import _hip

    def expect(ctype: str):
        t = ctype.replace("const ", "").strip()
        if t == "ssdseg_view*":
            return _hip._VP
        if t == "ssdseg_gview*":
            return _hip._GP
        if t == "int":
            return C.c_int
        if t == "float":
            return C.c_float
        if t == "double":
            return C.c_double
        if t == "size_t":
            return C.c_size_t
        if t == "long long":
            return C.c_longlong
        assert t.endswith("*"), t
        return "pointer"

    protos = _header_prototypes()
    assert len(protos) > 70
    for name, params in protos.items():
        if name == "ssdseg_version":
            continue
        sig = _hip._SIGNATURES[name]
        assert len(sig) == len(params), f"{name}: {len(sig)} ctypes parameters for {len(params)} C parameters"
        for i, (ct, p) in enumerate(zip(sig, params)):
            want = expect(p)
            if want == "pointer":
                assert ct in (C.c_void_p, C.c_char_p) or hasattr(ct, "contents") or issubclass(ct, C._Pointer), f"{name} arg {i + 1}: {p} bound as {ct}"
                assert ct not in (_hip._VP, _hip._GP), f"{name} arg {i + 1}: {p} bound as a view struct"
            else:
                assert ct is want, f"{name} arg {i + 1}: {p} bound as {ct}"


def test_bare_pointer_where_a_view_is_expected_is_a_python_error(lib):
    import ctypes as C
    from ssdseglib import _hip
    with pytest.raises(C.ArgumentError):
        # residual is `const ssdseg_view*` (4th parameter): a raw device pointer must not be accepted
        lib.ssdseg_bn_apply(None, C.byref(_hip.view(None)), 4, C.c_void_p(0x7f0000000000), 4, None, 4, 1, 4)
    with pytest.raises(C.ArgumentError):
        lib.ssdseg_pwconv_fwd(None, C.c_void_p(0x7f0000000000), 4, None, None, 4, 1, 4, 4, None)
    assert lib.ssdseg_bn_apply(None, C.byref(_hip.view(None)), 4, None, 4, None, 4, 1, 4) == -1001   # reaches C: ctx == NULL -> EINVAL(1)


# ---------------------------------------------------------------------------------------------- input pipeline, host side
def test_oracle_expand_inputs_and_flip_boxes_small_case():
    """the oracle's restatement of read_and_encode's tensor part (reference datacoder.py:325-345) on a case small enough to write
    down: cast, one-hot with an out-of-range index, left-right mirror of the flagged sample, x -> W - x for its boxes"""
    from oracle import np_ops as O
    img = np.arange(2 * 1 * 3 * 3, dtype=np.uint8).reshape(2, 1, 3, 3)
    idx = np.array([[[0, 2, 5]], [[1, 1, 0]]], np.uint8)
    out_img, out_mask = O.expand_inputs(img, idx, np.array([1, 0], np.uint8), 3)
    assert out_img.dtype == np.float32 and out_mask.dtype == np.float32
    np.testing.assert_array_equal(out_img[0, 0], [[6, 7, 8], [3, 4, 5], [0, 1, 2]])          # mirrored
    np.testing.assert_array_equal(out_img[1, 0], [[9, 10, 11], [12, 13, 14], [15, 16, 17]])  # untouched
    np.testing.assert_array_equal(out_mask[0, 0], [[0, 0, 0], [0, 0, 1], [1, 0, 0]])         # 5 -> zeros, mirrored
    np.testing.assert_array_equal(out_mask[1, 0], [[0, 1, 0], [0, 1, 0], [1, 0, 0]])
    gt = np.array([[2, 10, 20, 110, 220], [1, 0, 0, 639, 479]], np.float32)
    np.testing.assert_array_equal(O.flip_gt_boxes(gt, 640), [[2, 530, 20, 630, 220], [1, 1, 0, 640, 479]])   # W, not W - 1 (quirk Q8)
    np.testing.assert_array_equal(O.flip_gt_boxes(O.flip_gt_boxes(gt, 640), 640), gt)


def test_rgb_augmentation_against_colorsys(rng):
    """the host-side augmentation_rgb_channels (reference datacoder.py:452-464) with fixed draws vs a per-pixel colorsys
    restatement of tf.image.adjust_hue / adjust_saturation / adjust_contrast / adjust_brightness + clip"""
    import colorsys
    from ssdseglib import datacoder as D
    x = rng.integers(0, 256, (2, 5, 7, 3)).astype(np.float32)
    hue, sat, con, bri = 0.04, 1.05, 0.93, 0.08
    got = D._augment_rgb(x, hue, sat, con, bri)
    want = np.empty_like(x, dtype=np.float64)
    for i in np.ndindex(x.shape[:3]):
        h, s, v = colorsys.rgb_to_hsv(*(x[i].astype(np.float64) / 255.0))
        r, g, b = colorsys.hsv_to_rgb((h + hue) % 1.0, s, v)
        h, s, v = colorsys.rgb_to_hsv(r, g, b)
        want[i] = np.array(colorsys.hsv_to_rgb(h, min(max(s * sat, 0.0), 1.0), v)) * 255.0
    mean = want.mean(axis=(1, 2), keepdims=True)
    want = np.clip((want - mean) * con + mean + bri, 0.0, 255.0)
    np.testing.assert_allclose(got, want, atol=2e-3)
    # the public wrapper: bounded change, targets untouched, range kept
    targets = {"output-mask": object()}
    aug, t = D.augmentation_rgb_channels(x, targets)
    assert t is targets and aug.dtype == np.float32 and aug.min() >= 0.0 and aug.max() <= 255.0
    assert np.abs(aug - x).max() < 60.0


def test_no_register_soffset_store_data_hazard_in_the_built_library():
    """gfx950 needs two wait states between a buffer store of more than 64 bits and a VALU write of its data registers; hipcc
    inserts none when the store's soffset is an SGPR (round 2's wrong lanes 12-15 in csrc/pw_wgrad.h; pinned by
    scripts/micro/store_x4_hazard.hip, profiles/r03_store_x4_hazard.txt).  The built library must not contain the pattern."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(REPO, "scripts", "check_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from ssdseglib import _hip
    stores, findings = mod.scan(_hip.library_path())
    assert stores > 0, "the scan found no wide buffer store at all: has the disassembly format changed?"
    assert not findings, findings
