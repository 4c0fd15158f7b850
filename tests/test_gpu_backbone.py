"""GPU parity of the lowered MobileNetV2 backbone (stem, depthwise, pointwise, fused BN/ReLU6, residual adds):
forward activations and every parameter gradient vs the layer-by-layer NumPy oracle on the same weights/inputs."""
import numpy as np
import pytest

from oracle import np_ops as O
from oracle.np_model import NpModel

pytestmark = pytest.mark.gpu

TAPS = ['backbone-block16-project-batchnorm', 'backbone-block3-expand-relu6', 'backbone-block13-expand-relu6']


def build_backbone(shape):
    import ssdseglib
    from ssdseglib import _graph as K
    K.set_seed(7)
    dummy = np.zeros(4, np.float32)
    b = ssdseglib.models.MobileNetV2SsdSegBuilder(shape, 6, 4, dummy, dummy, dummy, dummy, (0.1, 0.1, 0.2, 0.2))
    inp = b._mobilenetv2_backbone()
    return K.Model(inputs=inp, outputs=[b._layers[n] for n in TAPS])


def device_relu_masks(eng, model):
    """derivative masks of every ReLU as the device sees them: z = scale*y + shift from the resident raw tensors"""
    masks = {}
    for l in model.layers:
        if type(l).__name__ != "ReLU":
            continue
        v = eng.vals[id(l.outputs[0])]
        s = v.store
        flat = s.buf.download().ravel()       # a concat slice starts `coff` floats into its parent: index rows explicitly
        y = flat[np.arange(s.m)[:, None] * s.ld + np.arange(s.c)[None, :]].astype(np.float64)
        z = y if v.scale is None else y * v.scale.download().astype(np.float64) + v.shift.download().astype(np.float64)
        z = z.astype(np.float32).reshape(s.n, s.h, s.w, s.c)[..., :l.outputs[0].shape[-1]]   # (drop zero-padded channels: ShuffleNetV2 '1x')
        masks[l.name] = O.act_mask(z, v.act)
    return masks


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("batch,shape", [(2, (80, 96, 3)), (3, (96, 128, 3))])
def test_backbone_forward_backward_parity(ctx, rng, kernel_family, batch, shape):
    from ssdseglib import _engine as E
    model = build_backbone(shape)
    # perturb BN affine so gamma/beta gradients and shifts are exercised
    for l in model.layers:
        if type(l).__name__ == "BatchNormalization":
            c = l.weights["gamma"].size
            l.weights["gamma"] = rng.uniform(0.7, 1.3, c).astype(np.float32)
            l.weights["beta"] = rng.normal(0, 0.3, c).astype(np.float32)
    eng = E.Engine(model, batch, training=True, ctx=ctx)
    x = rng.integers(0, 256, (batch,) + shape).astype(np.float32)
    ref = NpModel(model, dtype=np.float64)   # fp64 oracle: the deep BN stack amplifies fp32 rounding of EITHER side
    ref_out = ref.forward(x, training=True)

    eng.set_input(x)
    eng.forward()
    for i, name in enumerate(TAPS):
        got = eng.output(i)
        assert got.shape == ref_out[i].shape
        assert rel(got, ref_out[i]) < 1e-3, name

    gouts = [rng.normal(0, 1, o.shape).astype(np.float32).astype(np.float64) for o in ref_out]
    ref_grads = ref.backward(gouts, relu_masks=device_relu_masks(eng, model))
    for i, g in enumerate(gouts):
        eng.seed_output_grad(i, g)
    eng.backward_from_outputs()
    ctx.sync()
    worst = 0.0
    for l in model.layers:
        if not l.weights:
            continue
        # a beta feeding (1x1 conv -> training BN) has an exactly-zero true gradient: judge every tensor against the
        # largest gradient magnitude of its layer, not against its own (possibly pure-noise) magnitude
        scale = max(np.abs(ref_grads[l.name][w]).max() for w in l.trainable_names)
        for wname in l.trainable_names:
            got = eng.grad_view(l, wname).download()
            want = ref_grads[l.name][wname]
            err = np.abs(got.astype(np.float64) - want).max() / scale
            worst = max(worst, err)
            assert err < 2e-3, f"{l.name}/{wname}: rel err {err:.3e}"
    print("worst parameter-gradient rel err", worst)

    # moving statistics were updated like Keras does
    bn = model.get_layer('backbone-block1-expand-batchnorm')
    cache = ref.cache[bn.name]
    mm, mv = O.bn_moving_update(np.zeros_like(cache["mean"]), np.ones_like(cache["var"]), cache)
    got_mm, got_mv = bn.get_weights()[2:4]
    assert rel(got_mm, mm) < 1e-4 and rel(got_mv, mv) < 1e-4

    # second step on the same engine gives the same gradients (fixed-order reductions -> bit-identical)
    g1 = eng.P["grads"].download()
    eng.forward()
    for i, g in enumerate(gouts):
        eng.seed_output_grad(i, g)
    eng.backward_from_outputs()
    assert np.array_equal(g1, eng.P["grads"].download())


def test_deferred_column_sums_are_bit_identical_and_one_launch(ctx, rng, monkeypatch):
    """the column sums that fold weight-gradient partial slabs go out as ONE launch at the join that ends the
    backward pass (ssdseg_colsum_defer) instead of one per layer.  Same lanes, chains and fold order per column as the
    per-layer kernel: the gradient bucket is the same bits with deferral on and off; the kernel registry shows the launch counts."""
    from ssdseglib import _engine as E
    model = build_backbone((96, 128, 3))
    eng = E.Engine(model, 3, training=True, ctx=ctx)
    x = rng.integers(0, 256, (3, 96, 128, 3)).astype(np.float32)
    seeds = [rng.normal(0, 1, (3,) + tuple(t.shape[1:])).astype(np.float32) for t in model.outputs]

    def run(defer):
        monkeypatch.setenv("SSDSEG_COLSUM_DEFER", "1" if defer else "0")
        eng.set_input(x)
        eng.forward()
        for i, g in enumerate(seeds):
            eng.seed_output_grad(i, g)
        ctx.timing(True)
        ctx.timing_reset()
        eng.backward_from_outputs()
        rep = ctx.timing_report()
        ctx.timing(False)
        return eng.P["grads"].download(), rep

    g_off, rep_off = run(False)
    g_on, rep_on = run(True)
    g_off2, _ = run(False)
    assert np.isfinite(g_on).all() and np.abs(g_on).max() > 0
    assert np.array_equal(g_on, g_off) and np.array_equal(g_off, g_off2)
    assert rep_off.get("colsum_kernel", {"count": 0})["count"] > 20 and "colsum_batch_kernel" not in rep_off
    assert rep_on["colsum_batch_kernel"]["count"] == 1
    assert rep_on.get("colsum_kernel", {"count": 0})["count"] <= 4        # (consumed-at-once tables keep their own launch)
    ctx.colsum_defer(False)
