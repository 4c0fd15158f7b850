"""GPU parity of the input side of the step: a compact batch -- uint8 pixels, uint8 class indices, ground-truth
rows, flip flags -- expanded on the device (csrc/inputs.hip + ssdseg_encode_targets) vs the oracle's restatement of
DataEncoderDecoder.read_and_encode (reference datacoder.py:302-347).  Byte / index work -- pixels, one-hot rows, mirrored boxes,
matched labels -- is bit-exact; the encoded offsets hold a float32 log (datacoder.py:268-269) and are compared at 1e-5 like
tests/test_gpu_head_ops.py::test_encode_targets_exact."""
import numpy as np
import pytest

from oracle import np_ops as O
from tests.test_gpu_full_model import CW, SHAPE, build, make_targets

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,h,w,c,with_flip", [(3, 7, 13, 4, True), (2, 9, 16, 3, True), (1, 5, 5, 8, False), (4, 48, 64, 4, True)])
def test_expand_inputs_bit_exact(ctx, rng, b, h, w, c, with_flip):
    img = rng.integers(0, 256, (b, h, w, 3)).astype(np.uint8)
    idx = rng.integers(0, c + 2, (b, h, w)).astype(np.uint8)         # c, c+1: out of range -> all-zero one-hot rows
    flip = (np.arange(b) % 2 == 0).astype(np.uint8) if with_flip else None
    want_img, want_mask = O.expand_inputs(img, idx, flip, c)
    d_img, d_mask = ctx.empty((b, h, w, 3)), ctx.empty((b, h, w, c))
    d_flip = ctx.empty(b, np.uint8).upload(flip) if flip is not None else None
    ctx.call("ssdseg_expand_inputs", ctx.empty(img.shape, np.uint8).upload(img), ctx.empty(idx.shape, np.uint8).upload(idx), d_flip, d_img, d_mask,
             b, h, w, c)
    np.testing.assert_array_equal(d_img.download(), want_img)
    np.testing.assert_array_equal(d_mask.download(), want_mask)
    # either half alone
    d_img.zero_(); d_mask.zero_()
    ctx.call("ssdseg_expand_inputs", ctx.empty(img.shape, np.uint8).upload(img), None, d_flip, d_img, None, b, h, w, c)
    ctx.call("ssdseg_expand_inputs", None, ctx.empty(idx.shape, np.uint8).upload(idx), d_flip, None, d_mask, b, h, w, c)
    np.testing.assert_array_equal(d_img.download(), want_img)
    np.testing.assert_array_equal(d_mask.download(), want_mask)


def test_expand_inputs_rejects_bad_arguments(ctx):
    from ssdseglib import _hip as H
    buf = ctx.empty((1, 2, 2, 3))
    u8 = ctx.empty((1, 2, 2, 3), np.uint8)
    with pytest.raises(H.SsdsegError):
        ctx.call("ssdseg_expand_inputs", None, None, None, buf, None, 1, 2, 2, 4)       # nothing to expand
    with pytest.raises(H.SsdsegError):
        ctx.call("ssdseg_expand_inputs", u8, None, None, None, None, 1, 2, 2, 4)        # no destination
    with pytest.raises(H.SsdsegError):
        ctx.call("ssdseg_expand_inputs", u8, None, None, buf, None, 1, 2, 2, 9)         # more classes than the kernel's row


def test_flip_gt_boxes_bit_exact(ctx, rng):
    b, gmax, width = 5, 6, 640.0
    gt = np.zeros((b, gmax, 5), np.float32)
    cnt = rng.integers(0, gmax + 1, b).astype(np.int32)
    cnt[0] = gmax
    for n in range(b):
        x0 = rng.uniform(0, 500, cnt[n]); x1 = x0 + rng.uniform(1, 139, cnt[n])
        gt[n, :cnt[n]] = np.stack([rng.integers(1, 4, cnt[n]), x0, rng.uniform(0, 400, cnt[n]), x1, rng.uniform(400, 479, cnt[n])], axis=1)
    gt[:, :, 0][gt[:, :, 0] == 0] = 0
    flip = np.array([1, 0, 1, 1, 0], np.uint8)
    want = gt.copy()
    for n in range(b):
        if flip[n]:
            want[n, :cnt[n]] = O.flip_gt_boxes(gt[n, :cnt[n]], width)
    d = ctx.array(gt)
    ctx.call("ssdseg_flip_gt_boxes", d, ctx.empty(b, np.int32).upload(cnt), ctx.empty(b, np.uint8).upload(flip), b, gmax, width)
    np.testing.assert_array_equal(d.download(), want)       # rows past the count and unflagged samples untouched


def _compile(model):
    import ssdseglib
    model.compile(optimizer=ssdseglib.optimizers.Adam(learning_rate=1e-3),
                  loss={'output-mask': ssdseglib.losses.cross_entropy(classes_weights=CW), 'output-labels': ssdseglib.losses.confidence_loss,
                        'output-boxes': ssdseglib.losses.localization_loss},
                  loss_weights={'output-mask': 1.0, 'output-labels': 1.0, 'output-boxes': 1.0})


def _compact_batches(rng, sizes):
    """compact batches and, beside them, the float tensors the reference's map would hand over for the same samples, from the
    ORACLE (expand + flip + encode_targets restated in NumPy)"""
    import ssdseglib
    out = []
    for bsz in sizes:
        boxes, _, _ = build()
        enc, gts, targets = make_targets(rng, boxes, bsz)
        idx = targets['output-mask'].argmax(-1).astype(np.uint8)
        img = rng.integers(0, 256, (bsz,) + SHAPE).astype(np.uint8)
        flip = rng.integers(0, 2, bsz).astype(np.uint8)
        flip[0] = 1
        cb = ssdseglib.datacoder.CompactBatch(img, idx, gts, flip, enc)
        want_img, want_mask = O.expand_inputs(img, idx, flip, 4)
        corners = np.stack([enc.xmin_boxes_default, enc.ymin_boxes_default, enc.xmax_boxes_default, enc.ymax_boxes_default], axis=1).astype(np.float32)
        labels, offsets = [], []
        for g, f in zip(gts, flip):
            g = O.flip_gt_boxes(g, SHAPE[1]) if f else g
            l, o, _ = O.encode_targets(corners, g, 4, 0.525, enc._stds)
            labels.append(l); offsets.append(o)
        out.append((cb, want_img, {'output-mask': want_mask, 'output-labels': np.stack(labels), 'output-boxes': np.stack(offsets)}))
    return out


def _device_encoded(cb, targets):
    """the same float targets with the offsets from the DEVICE encoder on the oracle-mirrored boxes (the float32 log inside
    differs from NumPy's by an ulp now and then; a bit-for-bit comparison of two fits needs identical inputs)"""
    gts = [O.flip_gt_boxes(g, SHAPE[1]) if f else g for g, f in zip(cb.ground_truth, cb.flip)]
    labels, offsets = cb.encoder.encode_batch(gts)
    np.testing.assert_array_equal(labels, targets['output-labels'])
    return dict(targets, **{'output-boxes': offsets})


def test_compact_batch_fills_the_step_buffers_like_the_oracle(ctx, rng):
    """the engine's input / mask / label / offset buffers after a compact hand-over == the oracle's float tensors, bit for bit"""
    from ssdseglib import _engine as E
    E.set_default_context(ctx)
    (cb, want_img, want), = _compact_batches(rng, (3,))
    _, _, model = build(seed=5)
    _compile(model)
    eng = E.engine_for(model, 3, True)
    ld = E._compact_loader(eng, cb)
    ld.stage(cb)
    ld.consume()
    np.testing.assert_array_equal(eng.input_store.buf.download().reshape(want_img.shape), want_img)
    for name, op, kind in eng._loss_names:
        got = (op.y_true if kind == "mask" else (op.y_labels if kind == "conf" else op.y_boxes)).download()
        if kind == "loc":
            assert np.abs(got.reshape(want[name].shape) - want[name]).max() < 1e-5      # float32 log of the size ratio
        else:
            np.testing.assert_array_equal(got.reshape(want[name].shape), want[name], err_msg=name)


def test_fit_on_compact_batches_equals_fit_on_float_tensors(ctx, rng, monkeypatch):
    """model.fit over compact batches (uploads overlapped with the running step, and synchronous) gives the history of the
    same fit over the expanded float32 tensors -- identical device kernels on identical inputs, so identical bits; a smaller
    last batch goes through another engine"""
    from ssdseglib import _engine as E
    E.set_default_context(ctx)
    data = [(cb, img, _device_encoded(cb, t)) for cb, img, t in _compact_batches(rng, (3, 3, 3, 2))]
    hist = {}
    for mode in ("compact-overlap", "compact-sync", "float"):
        monkeypatch.setenv("SSDSEG_FIT_OVERLAP", "0" if mode == "compact-sync" else "1")
        _, _, model = build(seed=5)
        _compile(model)
        batches = [cb for cb, _, _ in data] if mode != "float" else [(img, t) for _, img, t in data]
        hist[mode] = model.fit(batches, epochs=2, verbose=0).history
    for mode in ("compact-overlap", "compact-sync"):
        assert hist[mode].keys() == hist["float"].keys()
        for k in hist["float"]:
            assert hist[mode][k] == hist["float"][k], (mode, k, hist[mode][k], hist["float"][k])
    # train_on_batch takes the same object
    _, _, model = build(seed=5)
    _compile(model)
    logs = model.train_on_batch(data[0][0])
    _, _, model2 = build(seed=5)
    _compile(model2)
    assert logs == model2.train_on_batch(data[0][1], data[0][2])


def test_compact_batch_errors(ctx, rng):
    import ssdseglib
    from ssdseglib import _engine as E
    E.set_default_context(ctx)
    (cb, _, _), = _compact_batches(rng, (2,))
    with pytest.raises(ValueError):
        ssdseglib.datacoder.CompactBatch(cb.images, cb.mask_index[:, :-1], cb.ground_truth, cb.flip, cb.encoder)
    with pytest.raises(ValueError):
        ssdseglib.datacoder.CompactBatch(cb.images, cb.mask_index, cb.ground_truth[:1], cb.flip, cb.encoder)
    _, _, model = build(seed=5)
    _compile(model)
    crowded = ssdseglib.datacoder.CompactBatch(cb.images, cb.mask_index, [np.tile(cb.ground_truth[0][:1], (65, 1)), cb.ground_truth[1]], cb.flip, cb.encoder)
    with pytest.raises(ValueError, match="64"):
        model.train_on_batch(crowded)
    # an encoder built for another class count than the model's heads: the encode kernel would overrun / mis-stride the target
    # buffers -- refused when the loader is made
    import copy
    enc5 = copy.copy(cb.encoder)
    enc5.num_classes = 5
    wrong = ssdseglib.datacoder.CompactBatch(cb.images, cb.mask_index, cb.ground_truth, cb.flip, enc5)
    _, _, model2 = build(seed=5)
    _compile(model2)
    with pytest.raises(ValueError, match="num_classes"):
        model2.train_on_batch(wrong)
