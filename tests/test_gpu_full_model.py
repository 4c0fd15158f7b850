"""GPU parity of the whole multi-task network (MobileNetV2 + DeepLabV3+ + SSDLite, reference models.py:314-343):
outputs, the three losses and every parameter gradient of one training step vs the NumPy oracle; the inference
model (decode + segmentation suppression + combined NMS, reference models.py:345-423) end to end; fit/predict API."""
import numpy as np
import pytest

from oracle import np_ops as O
from oracle.np_model import NpModel
from tests.test_gpu_backbone import device_relu_masks, rel

pytestmark = pytest.mark.gpu

SHAPE = (96, 128, 3)
FMAPS = ((6, 8), (3, 4), (2, 2), (1, 1))
CW = (0.05, 0.575, 0.135, 0.24)
STDS = (0.1, 0.1, 0.2, 0.2)


def build(seed=11):
    import ssdseglib
    from ssdseglib import _graph as K
    K.set_seed(seed)
    boxes = ssdseglib.boxes.DefaultBoundingBoxes(feature_maps_shapes=FMAPS, centers_padding_from_borders_percentage=0.05,
                                                 boxes_scales=(0.15, 0.95))
    boxes.rescale_boxes_coordinates(SHAPE[:2])
    builder = ssdseglib.models.MobileNetV2SsdSegBuilder(
        SHAPE, [6, 6, 6, 6], 4, boxes.get_boxes_coordinates_center_x('ssd'), boxes.get_boxes_coordinates_center_y('ssd'),
        boxes.get_boxes_coordinates_width('ssd'), boxes.get_boxes_coordinates_height('ssd'), STDS)
    model = builder.get_model_for_training('deeplabv3plus', 'ssdlite', (3, 6, 12))
    return boxes, builder, model


def make_targets(rng, boxes, batch):
    import ssdseglib
    enc = ssdseglib.datacoder.DataEncoderDecoder(
        4, SHAPE[:2], xmin_boxes_default=boxes.get_boxes_coordinates_xmin('ssd'), ymin_boxes_default=boxes.get_boxes_coordinates_ymin('ssd'),
        xmax_boxes_default=boxes.get_boxes_coordinates_xmax('ssd'), ymax_boxes_default=boxes.get_boxes_coordinates_ymax('ssd'),
        iou_threshold=0.525, standard_deviations_centroids_offsets=STDS)
    gts, mask = [], np.zeros((batch,) + SHAPE[:2], np.int64)
    for b in range(batch):
        g = int(rng.integers(1, 4))
        w = rng.uniform(20, 70, g); h = rng.uniform(20, 60, g)
        x0 = rng.uniform(0, SHAPE[1] - w - 1); y0 = rng.uniform(0, SHAPE[0] - h - 1)
        lab = rng.integers(1, 4, g)
        gts.append(np.stack([lab, x0, y0, x0 + w, y0 + h], axis=1).astype(np.float32))
        for l, xa, ya, ww, hh in zip(lab, x0, y0, w, h):
            mask[b, int(ya):int(ya + hh), int(xa):int(xa + ww)] = l
    labels, offsets = enc.encode_batch(gts)
    return enc, gts, {'output-mask': np.eye(4, dtype=np.float32)[mask], 'output-labels': labels, 'output-boxes': offsets}


@pytest.mark.parametrize("mask_loss", ["cross_entropy", "dice", "dice_square"])
def test_full_train_step_parity(ctx, rng, mask_loss):
    import ssdseglib
    from ssdseglib import _engine as E
    batch = 3       # (2 would make the 1x1-map BatchNorms see two samples, whose input gradient is identically zero)
    boxes, builder, model = build()
    for l in model.layers:
        if type(l).__name__ == "BatchNormalization":
            c = l.weights["gamma"].size
            l.weights["gamma"] = rng.uniform(0.7, 1.3, c).astype(np.float32)
            l.weights["beta"] = rng.normal(0, 0.3, c).astype(np.float32)
    enc, gts, targets = make_targets(rng, boxes, batch)
    # the encoder output agrees with the oracle's encoder (exact matching, close offsets)
    corners = boxes.get_boxes_coordinates_corners('ssd')
    for b in range(batch):
        l_ref, o_ref, _ = O.encode_targets(corners, gts[b], 4, 0.525, STDS)
        assert np.array_equal(targets['output-labels'][b], l_ref)
        assert np.abs(targets['output-boxes'][b] - o_ref).max() < 1e-5
    assert targets['output-labels'][..., 1:].sum() > 0, "test needs at least one positive anchor"

    model.compile(optimizer=ssdseglib.optimizers.Adam(learning_rate=1e-4),
                  loss={'output-mask': getattr(ssdseglib.losses, mask_loss)(classes_weights=CW), 'output-labels': ssdseglib.losses.confidence_loss,
                        'output-boxes': ssdseglib.losses.localization_loss},
                  loss_weights={'output-mask': 1.0, 'output-labels': 1.0, 'output-boxes': 1.0})
    E.Engine.keep_mask_probabilities = True
    try:
        eng = E.Engine(model, batch, training=True, ctx=ctx)
    finally:
        E.Engine.keep_mask_probabilities = False
    eng.configure_losses(model._compiled["loss"], model._compiled["loss_weights"])
    x = rng.integers(0, 256, (batch,) + SHAPE).astype(np.float32)

    ref = NpModel(model, dtype=np.float64)
    p_mask, p_labels, p_boxes = ref.forward(x, training=True)
    if mask_loss == "cross_entropy":
        l_mask, dmask = O.cross_entropy_loss(targets['output-mask'].astype(np.float64), p_mask, np.asarray(CW, np.float64))
    else:       # the reference's dice losses as the training loss of the mask head (losses.py:175-264)
        l_mask, dmask = O.dice_loss_grad(targets['output-mask'].astype(np.float64), p_mask, np.asarray(CW, np.float64), squared=mask_loss == "dice_square")
    l_conf, dconf, _ = O.confidence_loss(targets['output-labels'].astype(np.float64), p_labels)
    l_loc, dloc = O.localization_loss(targets['output-boxes'].astype(np.float64), p_boxes)

    eng.set_input(x)
    eng.set_targets(targets)
    eng.forward()
    assert np.abs(eng.output(0) - p_mask).max() < 2e-4
    assert np.abs(eng.output(1) - p_labels).max() < 2e-4
    assert rel(eng.output(2), p_boxes) < 1e-3
    got = eng.losses()
    assert abs(got['output-mask_loss'] - l_mask.mean()) < 1e-3 * abs(l_mask.mean())
    assert abs(got['output-labels_loss'] - l_conf.mean()) < 1e-3 * abs(l_conf.mean())
    assert abs(got['output-boxes_loss'] - l_loc.mean()) < 1e-3 * abs(l_loc.mean())
    assert abs(got['loss'] - (l_mask.mean() + l_conf.mean() + l_loc.mean())) < 1e-3 * got['loss']

    eng.backward()
    ctx.sync()
    ref_grads = ref.backward([dmask / batch, dconf / batch, dloc / batch], relu_masks=device_relu_masks(eng, model))
    worst, worst_name = 0.0, ""
    for l in model.layers:
        if not l.weights:
            continue
        scale = max(np.abs(ref_grads[l.name][w]).max() for w in l.trainable_names)
        if scale == 0:   # e.g. the 1x1-map heads when mining selects none of their 6 anchors: the gradient is exactly zero
            assert all(np.abs(eng.grad_view(l, w).download()).max() < 1e-12 for w in l.trainable_names), l.name
            continue
        for wname in l.trainable_names:
            g = eng.grad_view(l, wname).download()
            err = np.abs(g.astype(np.float64) - ref_grads[l.name][wname]).max() / scale
            if err > worst:
                worst, worst_name = err, f"{l.name}/{wname}"
            assert err < 1e-3, f"{l.name}/{wname}: rel err {err:.3e}"
    print("worst parameter-gradient rel err", worst, worst_name)

    # Adam step == Keras formula on the flat bucket
    p0, g0 = eng.P["params"].download(), eng.P["grads"].download()
    eng.adam_step(lr=1e-4)
    want, _, _ = O.adam_step(p0.astype(np.float64), g0.astype(np.float64), np.zeros_like(p0, np.float64), np.zeros_like(p0, np.float64), 1)
    assert np.abs(eng.P["params"].download() - want).max() < 1e-7


def test_inference_model_and_keras_surface(ctx, rng):
    import ssdseglib
    from ssdseglib import _engine as E
    E.set_default_context(ctx)
    boxes, builder, model = build(seed=5)
    enc, gts, targets = make_targets(rng, boxes, 3)
    model.compile(optimizer=ssdseglib.optimizers.Adam(learning_rate=1e-3),
                  loss={'output-mask': ssdseglib.losses.cross_entropy(classes_weights=CW), 'output-labels': ssdseglib.losses.confidence_loss,
                        'output-boxes': ssdseglib.losses.localization_loss})
    x = rng.integers(0, 256, (3,) + SHAPE).astype(np.float32)
    # fit with a partial last batch (2 + 1) and validation data; losses go down on the training batch
    ds = [(x[:2], {k: v[:2] for k, v in targets.items()}), (x[2:], {k: v[2:] for k, v in targets.items()})]
    hist = model.fit(ds, epochs=4, validation_data=ds[:1], verbose=0)
    assert set(hist.history) >= {'loss', 'output-mask_loss', 'output-labels_loss', 'output-boxes_loss', 'val_loss'}
    assert hist.history['loss'][-1] < hist.history['loss'][0]
    assert all(np.isfinite(v) for vs in hist.history.values() for v in vs)

    inference = builder.get_model_for_inference(model_trained=model, max_number_of_boxes_per_class=4, max_number_of_boxes_per_sample=10,
                                                boxes_iou_threshold=0.3, labels_probability_threshold=0.26, suppress_background_boxes=False,
                                                use_segmentation_suppression=True)
    seg, det = inference.predict([x[:2], x[2:]])
    assert seg.shape == (3,) + SHAPE[:2] + (4,) and det.shape == (3, 10, 6)
    # oracle on the same (trained, downloaded) weights in inference mode
    ref = NpModel(inference, dtype=np.float32)
    ref.set_weights_from(lambda l: l.get_weights())
    seg_ref, det_ref = ref.forward(x[:2], training=False)
    assert np.abs(seg[:2] - seg_ref).max() < 1e-3
    # detections: the index kernels are exact given the same inputs -> feed the oracle tail with the device tensors
    eng = E.engine_for(inference, 2, False)
    eng.set_input(x[:2]); eng.forward()
    probs = eng.vals[id(inference.get_layer('output-labels').outputs[0])].store.buf.download().reshape(2, -1, 4)
    offs = eng.vals[id(inference.get_layer('output-boxes').outputs[0])].store.buf.download().reshape(2, -1, 4)
    mask = eng.output(0)
    assert np.abs(probs - ref.value('output-labels')).max() < 1e-3
    dec = inference.get_layer('decode-output-boxes')
    cent = np.stack([dec.center_x_boxes_default, dec.center_y_boxes_default, dec.width_boxes_default, dec.height_boxes_default], axis=1)
    corners = O.decode_to_corners_pred(offs, cent, STDS)
    want, _ = O.combined_nms(corners, O.seg_suppress(mask, probs), 4, 10, 0.3, 0.26)
    got = eng.output(1)
    assert np.array_equal(got[..., 0], want[..., 0])                 # same classes in the same order
    assert np.abs(got - want).max() < 1e-3
    # call syntax of NB03#cell31 and the checkpoint round trip
    seg1, det1 = inference(x[:1], training=False)
    assert np.abs(seg1 - seg[:1]).max() < 1e-6 and np.array_equal(det1, det[:1])
    import os, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "ckpt.npz")
        model.save(path)
        _, _, fresh = build(seed=99)
        fresh.load_weights(path)
        for a, b in zip(model.layers, fresh.layers):
            for wa, wb in zip(a.get_weights(), b.get_weights()):
                assert np.array_equal(wa, wb)


def test_fit_overlapped_upload_equals_synchronous(ctx, rng, monkeypatch):
    """fit() stages batch i+1 on the copy stream while step i runs (pinned host buffers -> device staging -> the engine's
    buffers at the start of the next step): the history must be bit-identical to the synchronous hand-over, also across a
    smaller last batch (which takes the synchronous path: another engine)"""
    import ssdseglib
    from ssdseglib import _engine as E
    E.set_default_context(ctx)
    batches = []
    for bsz in (3, 3, 3, 2):
        boxes, builder, _ = build()
        enc, gts, targets = make_targets(rng, boxes, bsz)
        batches.append((rng.integers(0, 256, (bsz,) + SHAPE).astype(np.float32), targets))
    hist = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SSDSEG_FIT_OVERLAP", mode)
        boxes, builder, model = build(seed=5)
        model.compile(optimizer=ssdseglib.optimizers.Adam(learning_rate=1e-3),
                      loss={'output-mask': ssdseglib.losses.cross_entropy(classes_weights=CW), 'output-labels': ssdseglib.losses.confidence_loss,
                            'output-boxes': ssdseglib.losses.localization_loss},
                      loss_weights={'output-mask': 1.0, 'output-labels': 1.0, 'output-boxes': 1.0})
        hist[mode] = model.fit(batches, epochs=2, verbose=0).history
    assert hist["1"].keys() == hist["0"].keys()
    for k in hist["1"]:
        assert hist["1"][k] == hist["0"][k], (k, hist["1"][k], hist["0"][k])


def test_detection_branch_on_the_side_stream_is_bit_identical(ctx, rng, monkeypatch):
    """Round 3: a training engine issues the detection branch (extra feature maps, SSD heads, detection losses) on the
    context's side stream, beside the mask branch (Engine._schedule).  Same kernels, same backward issue order, hence the same
    first-writer / accumulate decisions and summation orders: gradients, losses, parameters and moving statistics are the
    same bits with everything on one stream (SSDSEG_DET_SIDE=0).  Also with metrics bound (they read both branches' outputs)."""
    import ssdseglib
    from ssdseglib import _engine as E
    batch = 3
    boxes, builder, model = build(seed=23)
    enc, gts, targets = make_targets(rng, boxes, batch)
    model.compile(optimizer=ssdseglib.optimizers.Adam(learning_rate=1e-3),
                  loss={'output-mask': ssdseglib.losses.cross_entropy(classes_weights=CW), 'output-labels': ssdseglib.losses.confidence_loss,
                        'output-boxes': ssdseglib.losses.localization_loss},
                  metrics={'output-labels': ssdseglib.metrics.categorical_accuracy(classes_weights=(1.0, 1.0, 1.0, 1.0))})
    eng = E.Engine(model, batch, training=True, ctx=ctx)
    eng.configure_losses(model._compiled["loss"], model._compiled["loss_weights"])
    eng.configure_metrics(model._compiled["metrics"])
    trunk, det, mask, join_before = eng._schedule()
    assert len(det) > 30 and len(mask) > 20 and len(trunk) > 50 and join_before, (len(trunk), len(det), len(mask), len(join_before))
    assert any(op.name == "det-loss" for op in det) and all("mask" in op.name or "output-mask" in op.name for op in mask)
    # block 13's depthwise conv only feeds the detection outputs, but it completes the block-13 tap's gradient (and takes that
    # BatchNorm's backward sums over the completed tensor): it stays with the trunk, behind the ASPP's contributions
    assert any(op.name == "backbone-block13-depthwise-conv:dw" for op in trunk)
    assert any(op.name.startswith("backbone-block14") for op in det) and not any(op.name.startswith("backbone-block12") for op in det)
    x = rng.integers(0, 256, (batch,) + SHAPE).astype(np.float32)
    P = eng.P
    p0, s0 = P["params"].download(), P["state"].download()

    def run(split):
        monkeypatch.setenv("SSDSEG_DET_SIDE", "1" if split else "0")
        P["params"].upload(p0); P["state"].upload(s0)
        P["adam_m"].zero_(); P["adam_v"].zero_(); P["step"] = 0
        eng.train_step(x, targets, optimizer=model._compiled["optimizer"])
        ctx.sync()
        return P["grads"].download(), P["params"].download(), P["state"].download(), eng.losses()

    a, b, c = run(True), run(False), run(True)
    for u, v in ((a, b), (a, c)):
        assert np.array_equal(u[0], v[0]) and np.array_equal(u[1], v[1]) and np.array_equal(u[2], v[2]) and u[3] == v[3]
    assert np.isfinite(a[0]).all() and np.abs(a[0]).max() > 0 and all(np.isfinite(v) for v in a[3].values())
