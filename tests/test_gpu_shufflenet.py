"""GPU parity of the ShuffleNetV2 variant (reference models.py:425-790): backbone forward + every parameter gradient vs the
oracle (max-pool, channel split / shuffle, ReLU-after-Add, biased stem), and the full model with the reference's quirk Q1
(heads built with ReLU(max_value=0.0): uniform probabilities, zero offsets, zero gradients)."""
import numpy as np
import pytest

from oracle.np_model import NpModel
from tests.test_gpu_backbone import device_relu_masks, rel

pytestmark = pytest.mark.gpu

TAPS = ['backbone-stage3-block7-reshape-post-channels-shuffle', 'backbone-stage4-block3-reshape-post-channels-shuffle',
        'backbone-stage2-block3-reshape-post-channels-shuffle']


def builder(shape, extra_dw, residual, size='0.5x'):
    import ssdseglib
    from ssdseglib import _graph as K
    K.set_seed(3)
    d = np.zeros(4, np.float32)
    return ssdseglib.models.ShuffleNetV2SsdSegBuilder(shape, size, extra_dw, residual, 6, 4, d, d, d, d, (0.1, 0.1, 0.2, 0.2))


@pytest.mark.parametrize("extra_dw,residual,size", [(True, True, '0.5x'), (False, False, '0.5x'),
                                                     (True, True, '1x'), (False, False, '1x'), (True, False, '2x')])
def test_shufflenet_backbone_parity(ctx, rng, extra_dw, residual, size):
    """'1x' / '2x': the stage-2 branches are 58 / 122 channels wide -- not whole 16-byte channel vectors; inside those units the
    engine works on zero-padded tensors and weights (SplitGatherOp / TableShuffleOp, padded parameter copies)"""
    from ssdseglib import _engine as E, _graph as K
    shape, batch = (96, 128, 3), 2
    b = builder(shape, extra_dw, residual, size)
    inp = b._shufflenetv2_backbone()
    model = K.Model(inputs=inp, outputs=[b._layers[n] for n in TAPS])
    for l in model.layers:
        if type(l).__name__ == "BatchNormalization":
            c = l.weights["gamma"].size
            l.weights["gamma"] = rng.uniform(0.7, 1.3, c).astype(np.float32)
            l.weights["beta"] = rng.normal(0, 0.3, c).astype(np.float32)
        if l.name == 'backbone-stage1-conv':
            l.weights["bias"] = rng.normal(0, 0.5, l.weights["bias"].shape).astype(np.float32)
    eng = E.Engine(model, batch, training=True, ctx=ctx)
    x = rng.integers(0, 256, (batch,) + shape).astype(np.float32)
    ref = NpModel(model, dtype=np.float64)
    ref_out = ref.forward(x, training=True)
    eng.set_input(x)
    eng.forward()
    for i, name in enumerate(TAPS):
        assert rel(eng.output(i), ref_out[i]) < 1e-3, name
    gouts = [rng.normal(0, 1, o.shape).astype(np.float32).astype(np.float64) for o in ref_out]
    for i, g in enumerate(gouts):
        eng.seed_output_grad(i, g)
    eng.backward_from_outputs()
    ctx.sync()
    ref_grads = ref.backward(gouts, relu_masks=device_relu_masks(eng, model))
    worst = 0.0
    for l in model.layers:
        if not l.weights:
            continue
        scale = max(np.abs(ref_grads[l.name][w]).max() for w in l.trainable_names)
        for wname in l.trainable_names:
            err = np.abs(eng.grad_array(l, wname).astype(np.float64) - ref_grads[l.name][wname]).max() / scale
            worst = max(worst, err)
            assert err < 1e-3, f"{l.name}/{wname}: {err:.3e}"
    print("worst parameter-gradient rel err", worst)
    # a second step reproduces the gradients bit for bit (slice gradient flags are reset, reductions are ordered)
    g1 = eng.P["grads"].download()
    eng.forward()
    for i, g in enumerate(gouts):
        eng.seed_output_grad(i, g)
    eng.backward_from_outputs()
    assert np.array_equal(g1, eng.P["grads"].download())


@pytest.mark.parametrize("size", ['0.5x', '1x'])
def test_shufflenet_full_model_quirk_q1(ctx, rng, size):
    import ssdseglib
    from ssdseglib import _engine as E
    E.set_default_context(ctx)
    shape = (96, 128, 3)
    b = builder(shape, True, True, size)
    model = b.get_model_for_training('deeplabv3plus', 'ssdlite', (3, 6, 12))
    x = rng.integers(0, 256, (2,) + shape).astype(np.float32)
    mask, labels, boxes = model(x, training=False)
    assert mask.shape == (2, 96, 128, 4) and labels.shape[0] == 2 and labels.shape[2] == 4
    assert np.allclose(mask, 0.25) and np.allclose(labels, 0.25) and np.all(boxes == 0)       # quirk Q1
    ref = NpModel(model, dtype=np.float32)
    r_mask, r_labels, r_boxes = ref.forward(x, training=False)
    assert np.allclose(mask, r_mask) and np.allclose(labels, r_labels) and np.array_equal(boxes, r_boxes)
    model.compile(optimizer=ssdseglib.optimizers.Adam(1e-4),
                  loss={'output-mask': ssdseglib.losses.cross_entropy((0.05, 0.575, 0.135, 0.24)),
                        'output-labels': ssdseglib.losses.confidence_loss, 'output-boxes': ssdseglib.losses.localization_loss})
    a = labels.shape[1]
    y_labels = np.zeros((2, a, 4), np.float32); y_labels[..., 0] = 1; y_labels[0, 3] = [0, 1, 0, 0]
    y_boxes = np.zeros((2, a, 4), np.float32); y_boxes[0, 3] = [0.5, -0.2, 1.0, 0.3]
    y_mask = np.eye(4, dtype=np.float32)[rng.integers(0, 4, (2, 96, 128))]
    logs = model.train_on_batch(x, {'output-mask': y_mask, 'output-labels': y_labels, 'output-boxes': y_boxes})
    assert np.isfinite(logs['loss']) and abs(logs['output-mask_loss'] - (-np.log(0.25) * (y_mask * np.array([0.05, 0.575, 0.135, 0.24])).sum((1, 2, 3))).mean()) < 1e-2


def test_copy2d_and_copy2d_batch_exact(ctx, rng):
    """ssdseg_copy2d / ssdseg_copy2d_batch (the packed <-> zero-padded weight copies of the odd-width ShuffleNetV2 units, reference
    models.py:557-603): a table of blocks with different shapes and leading dimensions in one launch == the copies one by one,
    bit for bit, and nothing outside the blocks is touched"""
    shapes = [(58, 58, 60, 58), (1, 58, 60, 58), (9, 58, 60, 58), (122, 122, 124, 122), (3, 7, 16, 9)]     # rows, cols, ldd, lds
    srcs = [rng.normal(0, 1, (r, lds)).astype(np.float32) for r, c, ldd, lds in shapes]
    d_src = [ctx.array(s) for s in srcs]
    sentinel = np.float32(-7.5)
    d_one = [ctx.array(np.full((r, ldd), sentinel, np.float32)) for r, c, ldd, lds in shapes]
    d_bat = [ctx.array(np.full((r, ldd), sentinel, np.float32)) for r, c, ldd, lds in shapes]
    for (r, c, ldd, lds), s, d in zip(shapes, d_src, d_one):
        ctx.call("ssdseg_copy2d", d, ldd, s, lds, r, c)
    table = ctx.array(np.asarray([(d.ptr, ldd, s.ptr, lds, r, c) for (r, c, ldd, lds), s, d in zip(shapes, d_src, d_bat)], dtype=np.int64))
    ctx.call("ssdseg_copy2d_batch", table, len(shapes), max(r * c for r, c, _, _ in shapes), sum(r * c for r, c, _, _ in shapes))
    for (r, c, ldd, lds), s, a, b in zip(shapes, srcs, d_one, d_bat):
        want = np.full((r, ldd), sentinel, np.float32)
        want[:, :c] = s[:, :c]
        assert np.array_equal(a.download(), want) and np.array_equal(b.download(), want)
