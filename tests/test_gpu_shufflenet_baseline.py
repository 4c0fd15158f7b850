"""ShuffleNetV2-only kernels AT THE BASELINE.json SIZE (configs[4]: batch 32 per GPU, 480x640; reference models.py:480-760).

tests/test_gpu_shufflenet.py compares whole backbones with the oracle at 96x128 / batch 2; the kernels bench.py times for
configs[4] run with other grids, chunkings and code paths at full size (two-pass max-pool with winner codes over 240x320x24,
table gathers over 153,600 rows of 116 -> 2 x 60 padded channels, the x8 mask-head backward that folds a 20x20 window by four
threads, the Winograd weight gradient with partial strips -- that one is `test_conv3x3_decoder_at_baseline_shape[60-80]`, the
whole step is `test_full_train_step_batch32_480x640_properties[shufflenet]`, both in test_gpu_baseline_shapes.py).  Images are
independent in all of these, so the device runs the FULL batch-32 tensor and the fp64 / exact oracle is evaluated on a few
whole images of it (first, last, one in the middle) -- every grid row / chunk boundary inside an image is covered, the batch
stride is covered by first vs last.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import np_ops as O

pytestmark = pytest.mark.gpu

IMAGES = (0, 13, 31)


def test_maxpool_stem_at_baseline_shape(ctx, monkeypatch):
    """models.py:629 after the biased stem conv + BN + ReLU: 32 x 240 x 320 x 24 -> 120 x 160 (SAME pads (0, 1)).  Forward exact;
    backward: winner-code form (default) == window scan bit for bit over the whole tensor, both == the oracle on whole images;
    ties (ReLU zeros are everywhere) go to the first window cell on both sides."""
    from ssdseglib import _hip as H
    n, h, w, c = 32, 240, 320, 24
    rng = np.random.default_rng(629)
    x = rng.standard_normal((n, h, w, c), dtype=np.float32)
    sc = rng.uniform(0.5, 1.5, c).astype(np.float32)
    sf = rng.normal(0, 0.5, c).astype(np.float32)
    a = np.maximum(x * sc + sf, np.float32(0))          # what the view yields, in float32 like the device
    v = H.view(ctx.array(x), ctx.array(sc), ctx.array(sf), O.ACT_RELU)
    ho, wo = 120, 160
    out = ctx.empty((n, ho, wo, c))
    ctx.call("ssdseg_maxpool3x3s2_fwd", v, out, n, h, w, c)
    got = out.download()
    for i in IMAGES:
        ref = O.maxpool3x3s2_fwd(a[i:i + 1])
        assert np.abs(got[i:i + 1] - ref).max() < 1e-6      # the view's fma vs mul+add: one rounding
    g = rng.standard_normal((n, ho, wo, c), dtype=np.float32)
    dg = ctx.array(g)
    d1, d2 = ctx.empty(x.shape), ctx.empty(x.shape)
    monkeypatch.delenv("SSDSEG_MAXPOOL_BWD", raising=False)
    ctx.call("ssdseg_maxpool3x3s2_bwd", v, dg, d1, n, h, w, c)
    monkeypatch.setenv("SSDSEG_MAXPOOL_BWD", "scan")
    ctx.call("ssdseg_maxpool3x3s2_bwd", v, dg, d2, n, h, w, c)
    monkeypatch.delenv("SSDSEG_MAXPOOL_BWD", raising=False)
    r1, r2 = d1.download(), d2.download()
    assert np.array_equal(r1, r2)
    # the oracle picks winners on ITS activated tensor; feed it the device-precision one so that ties are the same ties
    act_dev = ctx.empty(x.shape)
    ctx.call("ssdseg_channel_gather", v, c, act_dev, c, C.c_longlong(n * h * w), c, ctx.array(np.arange(c, dtype=np.int32)), 0)
    a_dev = act_dev.download()
    for i in IMAGES:
        ref = O.maxpool3x3s2_bwd(a_dev[i:i + 1], g[i:i + 1])
        assert np.abs(r1[i:i + 1] - ref).max() < 1e-5
    # every gradient lands somewhere: the sum over an image is preserved (size-independent property, all images)
    assert np.allclose(r1.reshape(n, -1).sum(axis=1, dtype=np.float64), g.reshape(n, -1).sum(axis=1, dtype=np.float64), atol=1e-2)


def split_tables(cin, widths, padded):
    """Split (models.py:573): forward table of part p (padded channel j <- input channel) and the backward table of the input"""
    fwd, bwd, off = [], [], 0
    for wd, pc in zip(widths, padded):
        fwd.append(np.array([off + i if i < wd else -1 for i in range(pc)], np.int32))
        bwd.append(np.array([j - off if off <= j < off + wd else -1 for j in range(cin)], np.int32))
        off += wd
    return fwd, bwd


def test_split_and_shuffle_tables_at_baseline_shape(ctx):
    """stage 2 of ShuffleNetV2 '1x' at batch 32: 32 x 60 x 80 x 116, Split into 58 | 58 (held as two zero-padded 60-channel
    tensors), channel shuffle (groups 2) of the concatenated padded branches back to 116 packed channels with the branches'
    lazily fused BatchNorm + ReLU applied in the same pass; gradients through the inverse tables.  Index work: exact."""
    from ssdseglib import _hip as H
    n, h, w, c = 32, 60, 80, 116
    m = n * h * w
    half, pad = 58, 60
    rng = np.random.default_rng(573)
    x = rng.standard_normal((m, c), dtype=np.float32)
    dx_ = ctx.array(x)
    tf, tb = split_tables(c, [half, half], [pad, pad])
    parts = [ctx.empty((m, pad)) for _ in range(2)]
    for p, t in zip(parts, tf):
        ctx.call("ssdseg_channel_gather", H.view(dx_), c, p, pad, C.c_longlong(m), pad, ctx.array(t), 0)
    got = [p.download() for p in parts]
    for k in range(2):
        assert np.array_equal(got[k][:, :half], x[:, k * half:(k + 1) * half]) and not got[k][:, half:].any()
    # Split backward: the parts' gradients gathered back, second part accumulating (+ a previous gradient in the slot)
    gparts = [rng.standard_normal((m, pad), dtype=np.float32) for _ in range(2)]
    base = rng.standard_normal((m, c), dtype=np.float32)
    gx = ctx.array(base)
    for k, (gp, t) in enumerate(zip(gparts, tb)):
        ctx.call("ssdseg_channel_gather", H.view(ctx.array(gp)), pad, gx, c, C.c_longlong(m), c, ctx.array(t), 1)
    want = base.copy()
    want[:, :half] += gparts[0][:, :half]
    want[:, half:] += gparts[1][:, :half]
    assert np.array_equal(gx.download(), want)

    # channel shuffle of concat(branch A, branch B) held as ONE 120-channel padded buffer with a per-channel BN + ReLU view
    cat = rng.standard_normal((m, 2 * pad), dtype=np.float32)
    cat[:, half:pad] = 0
    cat[:, pad + half:] = 0
    sc = rng.uniform(0.5, 1.5, 2 * pad).astype(np.float32)
    sf = rng.normal(0, 0.5, 2 * pad).astype(np.float32)
    phys_of = [off + i for off in (0, pad) for i in range(half)]              # logical channel -> physical column
    fwd = np.array([phys_of[(j % 2) * half + j // 2] for j in range(c)], np.int32)
    inv = np.full(2 * pad, -1, np.int32)
    inv[fwd] = np.arange(c, dtype=np.int32)
    out = ctx.empty((m, c))
    ctx.call("ssdseg_channel_gather", H.view(ctx.array(cat), ctx.array(sc), ctx.array(sf), O.ACT_RELU), 2 * pad, out, c, C.c_longlong(m), c,
             ctx.array(fwd), 0)
    logical = np.concatenate([cat[:, :half], cat[:, pad:pad + half]], axis=1)
    lsc, lsf = np.concatenate([sc[:half], sc[pad:pad + half]]), np.concatenate([sf[:half], sf[pad:pad + half]])
    ref = O.channel_shuffle(np.maximum(logical * lsc + lsf, np.float32(0)).reshape(n, h, w, c), 2).reshape(m, c)
    assert np.abs(out.download() - ref).max() < 1e-6
    ctx.call("ssdseg_channel_gather", H.view(ctx.array(logical)), c, out, c, C.c_longlong(m), c,
             ctx.array(np.array([(j % 2) * half + j // 2 for j in range(c)], np.int32)), 0)
    assert np.array_equal(out.download(), O.channel_shuffle(logical.reshape(n, h, w, c), 2).reshape(m, c))     # no view: exact
    # backward of the shuffle: gradient of the packed output scattered into the padded concat layout; padding columns get zero
    go = rng.standard_normal((m, c), dtype=np.float32)
    gcat = ctx.array(np.full((m, 2 * pad), 7.0, np.float32))
    ctx.call("ssdseg_channel_gather", H.view(ctx.array(go)), c, gcat, 2 * pad, C.c_longlong(m), 2 * pad, ctx.array(inv), 0)
    gc = gcat.download()
    want = np.zeros((m, 2 * pad), np.float32)
    want[:, fwd] = go
    assert np.array_equal(gc, want)
    # the whole-vector kernel (even halves: what '0.5x' / '1.5x' run) at the same row count
    c2 = 232
    x2 = rng.standard_normal((38400, c2), dtype=np.float32)
    o2 = ctx.empty(x2.shape)
    ctx.call("ssdseg_channel_shuffle", H.view(ctx.array(x2)), c2, o2, c2, 38400, c2, 2, 0)
    assert np.array_equal(o2.download(), O.channel_shuffle(x2.reshape(32, 30, 40, c2), 2).reshape(38400, c2))
    b2 = ctx.empty(x2.shape)
    ctx.call("ssdseg_channel_shuffle", H.view(o2), c2, b2, c2, 38400, c2, 2, 1)
    assert np.array_equal(b2.download(), x2)


@pytest.mark.parametrize("loss", ["cross_entropy", "dice"])
def test_mask_head_x8_at_baseline_shape(ctx, monkeypatch, loss):
    """ShuffleNetV2's decoder ends at 60 x 80 (models.py:748-758): the fused mask head up-samples x8 to 480 x 640.  Forward loss
    and the gradient w.r.t. the 60 x 80 x 4 logits at batch 32, vs the fp64 oracle on whole images (bilinear x8 -> softmax ->
    weighted cross-entropy / dice); the split-window tile kernel (default) vs the one-thread-per-pixel gather kernel on ALL."""
    n, h, w, c, f = 32, 60, 80, 4, 8
    rng = np.random.default_rng(748)
    logits = rng.normal(0, 2, (n, h, w, c)).astype(np.float32)
    cls = rng.integers(0, c, (n, h * f, w * f))
    y = np.eye(c, dtype=np.float32)[cls]
    cw = np.array([0.05, 0.575, 0.135, 0.24], np.float32)
    cwh = (C.c_float * 4)(*cw)
    dl, dy = ctx.array(logits), ctx.array(y)
    lossb, g = ctx.empty(n), ctx.empty(logits.shape)
    scale = 1.0 / n
    if loss == "cross_entropy":
        ctx.call("ssdseg_mask_head_fwd", dl, n, h, w, c, f, f, dy, cwh, None, lossb)
        ctx.call("ssdseg_mask_head_bwd", dl, n, h, w, c, f, f, dy, cwh, scale, g)
    else:
        coef = ctx.empty((n, 8))
        ctx.call("ssdseg_mask_head_fwd_dice", dl, n, h, w, c, f, f, dy, cwh, 0, None, lossb, coef)
        ctx.call("ssdseg_mask_head_bwd_dice", dl, n, h, w, c, f, f, dy, coef, 0, scale, g)
    got_l, got_g = lossb.download(), g.download()
    assert np.isfinite(got_g).all()
    for i in IMAGES:
        p = O.softmax(O.bilinear_fwd(logits[i:i + 1].astype(np.float64), f, f))
        if loss == "cross_entropy":
            l_ref, dp = O.cross_entropy_loss(y[i:i + 1].astype(np.float64), p, cw.astype(np.float64))
        else:
            l_ref, dp = O.dice_loss_grad(y[i:i + 1].astype(np.float64), p, cw.astype(np.float64), squared=False)
        g_ref = O.bilinear_bwd(O.softmax_bwd(p, dp * scale), f, f)
        assert abs(got_l[i] - l_ref[0]) < 1e-5 * abs(l_ref[0]), (i, got_l[i], l_ref)
        assert np.abs(got_g[i] - g_ref[0]).max() < 2e-5 * np.abs(g_ref).max(), i
    monkeypatch.setenv("SSDSEG_MASK_BWD", "gather")
    if loss == "cross_entropy":
        ctx.call("ssdseg_mask_head_bwd", dl, n, h, w, c, f, f, dy, cwh, scale, g)
    else:
        ctx.call("ssdseg_mask_head_bwd_dice", dl, n, h, w, c, f, f, dy, coef, 0, scale, g)
    alt = g.download()
    assert np.abs(alt - got_g).max() < 2e-6 * np.abs(got_g).max()      # same terms, the tile kernel folds the window in four parts
