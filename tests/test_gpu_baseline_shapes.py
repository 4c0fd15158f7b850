"""GPU parity AT THE BASELINE.json LAYER SHAPES (batch 32, 480x640), under the DEFAULT dispatch.

The dispatch of the C library is size-dependent (register-limited `OCC=1` GEMM instantiations from 150,000 rows, the
weights-resident / fused dx+dW kernels from 500,000 rows, row-chunked depthwise marches, the weight-gradient split regime),
so the instantiations bench.py times only exist at these sizes.  A full fp64 oracle of a 2.4M-row layer would take minutes
of NumPy, so every case compares
  * a random SAMPLE of output rows / pixels with the fp64 oracle evaluated only there (incl. first / last rows, image borders),
  * full-tensor reductions (BatchNorm partial sums, weight gradients) either with the fp64 reduction of the DEVICE output
    (checks the reduction epilogue; the output itself is pinned by the sample) or, where linear in the inputs, with the
    oracle's closed form (column sums),
  * weight gradients: exactly computed random entries plus random projections u^T dW v (linear in cheap per-row dots).
Tolerances as in test_gpu_conv_ops.py (2e-5 of the tensor's magnitude for single ops, 1e-4 for multi-million-element sums).
The last test is one whole batch-32 480x640 train step with size-independent properties: everything finite, two steps from the
same state bit-identical, encoder output and hard-negative mask equal to the oracle's on the same device tensors.
"""
import os

import numpy as np
import pytest

from oracle import np_ops as O

pytestmark = pytest.mark.gpu

RELU6 = O.ACT_RELU6


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def randn32(rng, shape, std=1.0):
    a = rng.standard_normal(shape, dtype=np.float32)
    if std != 1.0:
        a *= np.float32(std)
    return a


def off_threshold(x, scale, shift):
    """nudge raw values whose pre-activation sits within 1e-3 of a ReLU6 threshold (mask flips there are rounding noise)"""
    z = x * scale + shift
    bad = (np.abs(z) < 1e-3) | (np.abs(z - 6) < 1e-3)
    x[bad] += np.float32(0.05)
    return x


def view_inputs(rng, shape):
    c = shape[-1]
    scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
    shift = rng.uniform(-1, 3, c).astype(np.float32)
    x = off_threshold(randn32(rng, shape, 2.0), scale, shift)
    return x, scale, shift


def act64(x, scale, shift):
    return np.clip(x.astype(np.float64) * scale + shift, 0.0, 6.0)


def gview_inputs(rng, shape):
    c = shape[-1]
    g = randn32(rng, shape)
    scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
    shift = rng.uniform(-1, 3, c).astype(np.float32)
    y = off_threshold(randn32(rng, shape, 2.0), scale, shift)
    k1 = rng.normal(0, 0.1, c).astype(np.float32)
    k0 = rng.normal(0, 0.1, c).astype(np.float32)
    return g, y, scale, shift, k1, k0


def dy64(g, y, scale, shift, k1, k0):
    """the gradient view exactly as the kernels must form it, fp64, on whatever slice is passed"""
    z = y.astype(np.float64) * scale + shift
    return scale * ((z > 0) & (z < 6)) * g.astype(np.float64) + k1 * y.astype(np.float64) + k0.astype(np.float64)


def sample_rows(rng, m, count):
    idx = np.unique(np.concatenate([rng.integers(0, m, count), [0, 1, 127, 128, m - 129, m - 128, m - 2, m - 1]]))
    return idx[(idx >= 0) & (idx < m)]


# ------------------------------------------------------------------------------------------------ pointwise (K5)
PW_BASELINE = [
    # m, k, n                 layer
    (2457600, 16, 96),      # block-1 expand at 240x320: weights-resident fwd, fused dx+dW backward
    (2457600, 32, 16),      # block-0 project
    (614400, 256, 256),     # decoder sepconv pointwise at 120x160: OCC=1 four-tile kernels, long-M weight-gradient regime
    (614400, 144, 48),      # decoder 1x1 on the backbone tap
    (153600, 192, 32),      # 60x80 stage project (just above the OCC threshold)
    (38400, 576, 256),      # ASPP 1x1 branches at 30x40
    (38400, 1280, 256),     # ASPP output conv
    (38400, 96, 576),       # 30x40 expand
]


@pytest.mark.parametrize("m,k,n", PW_BASELINE)
def test_pointwise_at_baseline_shape(ctx, m, k, n):
    from ssdseglib import _hip as H
    rng = np.random.default_rng(m + k + n)
    x, sc, sh = view_inputs(rng, (m, k))
    wgt = (rng.normal(0, 1, (k, n)) / np.sqrt(k)).astype(np.float32)
    w64 = wgt.astype(np.float64)
    dx_, dsc, dsh, dw_ = ctx.array(x), ctx.array(sc), ctx.array(sh), ctx.array(wgt)
    dy_ = ctx.empty((m, n))
    nparts = ctx.parts("ssdseg_pwconv_parts", m, n)
    stats = ctx.empty((nparts, 2, n))
    ctx.call("ssdseg_pwconv_fwd", H.view(dx_, dsc, dsh, RELU6), k, dw_, dy_, n, m, k, n, stats)
    y = dy_.download()
    rows = sample_rows(rng, m, 4096)
    a_s = act64(x[rows], sc, sh)
    y_s = a_s @ w64
    scale_y = np.abs(y_s).max()
    assert np.abs(y[rows] - y_s).max() < 2e-5 * scale_y
    # BatchNorm partial sums: (a) the reduction epilogue vs the fp64 sums of the device output, (b) sum(y) vs the oracle's
    # closed form colsum(a) @ W (linear), over ALL rows
    st = stats.download().astype(np.float64).sum(axis=0)
    y64 = y.astype(np.float64)
    assert np.abs(st[0] - y64.sum(axis=0)).max() < 1e-4 * np.abs(y64).sum(axis=0).max()
    assert rel(st[1], (y64 ** 2).sum(axis=0)) < 1e-4
    colsum_a = np.zeros(k)
    for lo in range(0, m, 1 << 18):
        colsum_a += act64(x[lo:lo + (1 << 18)], sc, sh).sum(axis=0)
    assert np.abs(st[0] - colsum_a @ w64).max() < 1e-4 * np.abs(y64).sum(axis=0).max()
    del y64

    # backward: dx on sampled rows, dW by exact entries + random projections
    g, yraw, gs, gt, k1, k0 = gview_inputs(rng, (m, n))
    bufs = [ctx.array(v) for v in (g, yraw, gs, gt, k1, k0)]
    gv = H.gview(*bufs, act=RELU6)
    ddx, ddw = ctx.empty((m, k)), ctx.empty((k, n))
    res = randn32(rng, (m, k))
    dres = ctx.array(res)
    # dx + dW in one call: the gradient view carries the conv's OWN forward output (contract of ssdseg_pwconv_bwd: the fused
    # kernel of the big expand convs recomputes y = view(in) * w instead of reading it); the incoming gradient is zeroed where
    # the pre-activation sits within 1e-4 of a threshold (a last-bit difference of a recomputed y must not flip a mask that matters)
    near = np.zeros((m, n), bool)
    for lo in range(0, m, 1 << 18):
        z_own = y[lo:lo + (1 << 18)].astype(np.float64) * gs + gt
        near[lo:lo + (1 << 18)] = (np.abs(z_own) < 1e-4) | (np.abs(z_own - 6) < 1e-4)
    g_own = np.where(near, np.float32(0), g)
    del near
    gv_own = H.gview(ctx.array(g_own), dy_, *bufs[2:], act=RELU6)
    ctx.call("ssdseg_pwconv_bwd", H.view(dx_, dsc, dsh, RELU6), k, gv_own, n, dw_, ddx, k, ddw, m, k, n, dres, k, 0)
    dxg = ddx.download()
    dx_o = dy64(g_own[rows], y[rows], gs, gt, k1, k0) @ w64.T + res[rows]
    assert np.abs(dxg[rows] - dx_o).max() < 2e-5 * np.abs(dx_o).max()
    assert np.isfinite(dxg).all()
    check_wgrad(rng, ddw.download().astype(np.float64), lambda lo, hi: act64(x[lo:hi], sc, sh),
                lambda lo, hi: dy64(g_own[lo:hi], y[lo:hi], gs, gt, k1, k0), m)
    del g_own, gv_own
    dy_s = dy64(g[rows], yraw[rows], gs, gt, k1, k0)
    dx_s = dy_s @ w64.T + res[rows]
    # the separate kernels (what the engine calls when dx is not wanted / for the BN-fused variant): they read whatever y the view holds
    ctx.call("ssdseg_pwconv_bwd_weight", H.view(dx_, dsc, dsh, RELU6), k, gv, n, ddw, m, k, n)
    dwg = ddw.download().astype(np.float64)
    check_wgrad(rng, dwg, lambda lo, hi: act64(x[lo:hi], sc, sh), lambda lo, hi: dy64(g[lo:hi], yraw[lo:hi], gs, gt, k1, k0), m)
    ctx.call("ssdseg_pwconv_bwd_data", gv, n, dw_, ddx, k, m, k, n, None, 0, 0)
    assert np.abs(ddx.download()[rows] - (dx_s - res[rows])).max() < 2e-5 * np.abs(dx_s).max()
    if k > n:
        # project convs: backward-data with the producer BatchNorm's backward sums fused into the epilogue
        mean = np.zeros(k, np.float32)
        invstd = np.ones(k, np.float32)
        outs = [ctx.empty(k) for _ in range(4)]
        ctx.call("ssdseg_pwconv_bwd_bn", H.view(dx_, dsc, dsh, RELU6), k, gv, n, dw_, ddx, k, ddw, m, k, n, ctx.array(mean), ctx.array(invstd), *outs)
        dxb = ddx.download()
        assert np.abs(dxb[rows] - (dx_s - res[rows])).max() < 2e-5 * np.abs(dx_s).max()
        dbeta = np.zeros(k)
        dgamma = np.zeros(k)
        for lo in range(0, m, 1 << 18):
            xs = x[lo:lo + (1 << 18)].astype(np.float64)
            z = xs * sc + sh
            mg = dxb[lo:lo + (1 << 18)].astype(np.float64) * ((z > 0) & (z < 6))
            dbeta += mg.sum(axis=0)
            dgamma += (mg * xs).sum(axis=0)       # xhat == x for mean 0, invstd 1
        tol = 1e-4 * max(np.abs(dgamma).max(), np.abs(dbeta).max())
        assert np.abs(outs[0].download() - dgamma).max() < tol and np.abs(outs[1].download() - dbeta).max() < tol


def check_wgrad(rng, dw, a_rows, dy_rows, m, taps=None, chunk=1 << 17):
    """dw[k][n] = sum_m a[m][k] dy[m][n] checked over ALL rows: 12 exactly computed entries + 3 random projections u^T dW v,
    accumulated chunk-wise in fp64.  a_rows(lo, hi) / dy_rows(lo, hi) yield fp64 row blocks."""
    k, n = dw.shape
    ent = [(int(rng.integers(0, k)), int(rng.integers(0, n))) for _ in range(10)] + [(0, 0), (k - 1, n - 1)]
    us = rng.normal(0, 1, (3, k)); vs = rng.normal(0, 1, (3, n))
    e_acc = np.zeros(len(ent)); p_acc = np.zeros(3); mag = 0.0
    for lo in range(0, m, chunk):
        a, d = a_rows(lo, min(lo + chunk, m)), dy_rows(lo, min(lo + chunk, m))
        for i, (kk, nn) in enumerate(ent):
            e_acc[i] += a[:, kk] @ d[:, nn]
        au, dv = a @ us.T, d @ vs.T
        p_acc += (au * dv).sum(axis=0)
        mag += np.abs(a[:, ent[0][0]] * d[:, ent[0][1]]).sum()
    got_e = np.array([dw[kk, nn] for kk, nn in ent])
    assert np.abs(got_e - e_acc).max() < 5e-5 * max(np.abs(dw).max(), 1e-30), (got_e, e_acc)
    got_p = np.einsum("ik,kn,in->i", us, dw, vs)
    assert np.abs(got_p - p_acc).max() < 5e-5 * np.abs(us).sum(axis=1).max() * np.abs(dw).max() * np.sqrt(n), (got_p, p_acc)


# ------------------------------------------------------------------------------------------------ depthwise (K3)
DW_BASELINE = [
    # n, h, w, c, stride       layer
    (32, 240, 320, 96, 2),   # block-1 depthwise, the largest stride-2 layer (1180 MB forward)
    (32, 240, 320, 32, 1),   # block-0 depthwise
    (32, 120, 160, 256, 1),  # decoder sepconv depthwise (1259 MB forward)
    (32, 120, 160, 144, 1),  # block-2 depthwise: 144 channels = three 48-channel chunks x five strips per block (round 3)
    (32, 120, 160, 144, 2),  # block-3 depthwise
    (32, 30, 40, 576, 1),    # block-11/12 and the SSD head depthwise convs
]


def dw_fwd_at(a_fn, wgt, pix, s, pt, pl, h, w):
    """oracle depthwise output at sampled (n, ho, wo): sum_taps a[n, s*ho+kh-pt, s*wo+kw-pl, :] * wgt[kh, kw, :] (fp64)"""
    out = np.zeros((len(pix), wgt.shape[-1]))
    for kh in range(3):
        for kw in range(3):
            hh, ww = s * pix[:, 1] + kh - pt, s * pix[:, 2] + kw - pl
            ok = (hh >= 0) & (hh < h) & (ww >= 0) & (ww < w)
            out[ok] += a_fn(pix[ok, 0], hh[ok], ww[ok]) * wgt[kh, kw].astype(np.float64)
    return out


def dw_bwd_at(dy_fn, wgt, pix, s, pt, pl, ho, wo):
    """oracle dx at sampled input pixels (n, h, w): sum over taps with (h + pt - kh) % s == 0 of dy[n, (h+pt-kh)/s, ...] * w"""
    out = np.zeros((len(pix), wgt.shape[-1]))
    for kh in range(3):
        for kw in range(3):
            nh, nw = pix[:, 1] + pt - kh, pix[:, 2] + pl - kw
            ok = (nh % s == 0) & (nw % s == 0) & (nh >= 0) & (nw >= 0) & (nh // s < ho) & (nw // s < wo)
            out[ok] += dy_fn(pix[ok, 0], nh[ok] // s, nw[ok] // s) * wgt[kh, kw].astype(np.float64)
    return out


def sample_pixels(rng, n, h, w, count):
    pix = np.stack([rng.integers(0, n, count), rng.integers(0, h, count), rng.integers(0, w, count)], axis=1)
    edge = [(0, 0, 0), (0, 0, w - 1), (0, h - 1, 0), (n - 1, h - 1, w - 1), (n - 1, 0, w // 2), (n // 2, h - 1, w // 2), (n // 2, h // 2, 0)]
    return np.concatenate([pix, np.array(edge)], axis=0)


@pytest.mark.parametrize("n,h,w,c,s", DW_BASELINE)
def test_depthwise_at_baseline_shape(ctx, n, h, w, c, s):
    from ssdseglib import _hip as H
    rng = np.random.default_rng(h * w + c + s)
    x, sc, sh = view_inputs(rng, (n, h, w, c))
    wgt = rng.normal(0, 0.3, (3, 3, c)).astype(np.float32)
    ho, pt, _ = O.same_pad(h, 3, s)
    wo, pl, _ = O.same_pad(w, 3, s)
    dx_, dsc, dsh, dw_ = ctx.array(x), ctx.array(sc), ctx.array(sh), ctx.array(wgt)
    dy_ = ctx.empty((n, ho, wo, c))
    nparts = ctx.parts("ssdseg_dwconv_parts", n, h, w, c, s, 1)
    stats = ctx.empty((nparts, 2, c))
    ctx.call("ssdseg_dwconv_fwd", H.view(dx_, dsc, dsh, RELU6), dw_, dy_, n, h, w, c, s, 1, stats)
    y = dy_.download()
    a_fn = lambda nn, hh, ww: act64(x[nn, hh, ww], sc, sh)
    opix = sample_pixels(rng, n, ho, wo, 3000)
    y_s = dw_fwd_at(a_fn, wgt, opix, s, pt, pl, h, w)
    assert np.abs(y[opix[:, 0], opix[:, 1], opix[:, 2]] - y_s).max() < 2e-5 * np.abs(y_s).max()
    st = stats.download().astype(np.float64).sum(axis=0)
    ysum = np.zeros(c); ysq = np.zeros(c); yabs = np.zeros(c)
    for i in range(n):
        yi = y[i].astype(np.float64)
        ysum += yi.sum(axis=(0, 1)); ysq += (yi ** 2).sum(axis=(0, 1)); yabs += np.abs(yi).sum(axis=(0, 1))
    assert np.abs(st[0] - ysum).max() < 1e-4 * yabs.max()
    assert rel(st[1], ysq) < 1e-4
    del y

    g, yraw, gs, gt, k1, k0 = gview_inputs(rng, (n, ho, wo, c))
    bufs = [ctx.array(v) for v in (g, yraw, gs, gt, k1, k0)]
    gv = H.gview(*bufs, act=RELU6)
    ddx, ddw = ctx.empty(x.shape), ctx.empty(wgt.shape)
    mean = np.zeros(c, np.float32)
    invstd = np.ones(c, np.float32)
    outs = [ctx.empty(c) for _ in range(4)]
    # the variant the backbone runs: dx + dW + the producer BatchNorm's backward sums in one march
    ctx.call("ssdseg_dwconv_bwd_bn", H.view(dx_, dsc, dsh, RELU6), dw_, gv, ddx, ddw, n, h, w, c, s, 1, 0, ctx.array(mean), ctx.array(invstd), *outs)
    dxg = ddx.download()
    dy_fn = lambda nn, hh, ww: dy64(g[nn, hh, ww], yraw[nn, hh, ww], gs, gt, k1, k0)
    ipix = sample_pixels(rng, n, h, w, 3000)
    dx_s = dw_bwd_at(dy_fn, wgt, ipix, s, pt, pl, ho, wo)
    assert np.abs(dxg[ipix[:, 0], ipix[:, 1], ipix[:, 2]] - dx_s).max() < 2e-5 * max(np.abs(dx_s).max(), 1e-30)
    # dW[kh][kw][c] = sum a[n, s*ho+kh-pt, s*wo+kw-pl, c] * dy[n, ho, wo, c]: full fp64 reduction, image by image
    dw_ref = np.zeros((3, 3, c))
    dbeta = np.zeros(c); dgamma = np.zeros(c)
    for i in range(n):
        ai = np.zeros((h + 2, w + 2, c))
        ai[pt:pt + h, pl:pl + w] = act64(x[i], sc, sh)
        dyi = dy64(g[i], yraw[i], gs, gt, k1, k0)
        for kh in range(3):
            for kw in range(3):
                dw_ref[kh, kw] += (ai[kh:kh + (ho - 1) * s + 1:s, kw:kw + (wo - 1) * s + 1:s] * dyi).sum(axis=(0, 1))
        xi = x[i].astype(np.float64)
        z = xi * sc + sh
        mg = dxg[i].astype(np.float64) * ((z > 0) & (z < 6))
        dbeta += mg.sum(axis=(0, 1)); dgamma += (mg * xi).sum(axis=(0, 1))
    assert rel(ddw.download(), dw_ref) < 1e-4
    tol = 1e-4 * max(np.abs(dgamma).max(), np.abs(dbeta).max())
    assert np.abs(outs[0].download() - dgamma).max() < tol and np.abs(outs[1].download() - dbeta).max() < tol
    # plain backward with accumulation into an existing gradient (fan-out taps), same sample
    base = randn32(rng, (n, h, w, c))
    ddx.upload(base)
    ctx.call("ssdseg_dwconv_bwd", H.view(dx_, dsc, dsh, RELU6), dw_, gv, ddx, ddw, n, h, w, c, s, 1, 1)
    got = ddx.download()[ipix[:, 0], ipix[:, 1], ipix[:, 2]]
    assert np.abs(got - (dx_s + base[ipix[:, 0], ipix[:, 1], ipix[:, 2]])).max() < 2e-5 * max(np.abs(dx_s).max(), 1.0)
    assert rel(ddw.download(), dw_ref) < 1e-4


# ------------------------------------------------------------------------------------------------ dense 3x3 (K6)
@pytest.mark.parametrize("h,w", [(120, 160), (60, 80)])
def test_conv3x3_decoder_at_baseline_shape(ctx, h, w):
    """blocks.py:117 at batch 32: 32x120x160x304 -> 256 (MobileNetV2, 860.7 GFLOP per direction) and 32x60x80x304 -> 256
    (ShuffleNetV2: the decoder taps stage 2 at 60x80, models.py:748 -- the Winograd weight gradient then runs rows whose last
    32-column strip is partial: 80 = 2 * 32 + 16)"""
    from ssdseglib import _hip as H
    n, cin, cout = 32, 304, 256
    m = n * h * w
    rng = np.random.default_rng(304256 + h)
    x, sc, sh = view_inputs(rng, (n, h, w, cin))
    wgt = (rng.normal(0, 1, (3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    w64 = wgt.astype(np.float64)
    dx_, dsc, dsh, dw_ = ctx.array(x), ctx.array(sc), ctx.array(sh), ctx.array(wgt)
    dy_ = ctx.empty((n, h, w, cout))
    nparts = ctx.parts("ssdseg_conv3x3_parts", n, h, w, cin, cout)
    stats = ctx.empty((nparts, 2, cout))
    ctx.call("ssdseg_conv3x3_fwd", H.view(dx_, dsc, dsh, RELU6), cin, dw_, dy_, n, h, w, cin, cout, stats)
    y = dy_.download()
    pix = sample_pixels(rng, n, h, w, 1500)

    def conv_at(src_fn, wt, sign):
        out = np.zeros((len(pix), wt.shape[-1]))
        for kh in range(3):
            for kw in range(3):
                hh, ww = pix[:, 1] + sign * (kh - 1), pix[:, 2] + sign * (kw - 1)
                ok = (hh >= 0) & (hh < h) & (ww >= 0) & (ww < w)
                out[ok] += src_fn(pix[ok, 0], hh[ok], ww[ok]) @ wt[kh, kw]
        return out

    y_s = conv_at(lambda nn, hh, ww: act64(x[nn, hh, ww], sc, sh), w64, +1)
    assert np.abs(y[pix[:, 0], pix[:, 1], pix[:, 2]] - y_s).max() < 2e-5 * np.abs(y_s).max()
    st = stats.download().astype(np.float64).sum(axis=0)
    ysum = np.zeros(cout); ysq = np.zeros(cout); yabs = np.zeros(cout)
    for i in range(n):
        yi = y[i].astype(np.float64)
        ysum += yi.sum(axis=(0, 1)); ysq += (yi ** 2).sum(axis=(0, 1)); yabs += np.abs(yi).sum(axis=(0, 1))
    assert np.abs(st[0] - ysum).max() < 1e-4 * yabs.max()
    assert rel(st[1], ysq) < 1e-4
    del y

    # backward: the engine materialises the BatchNorm gradient view first (nine taps would each re-form it), then both kernels
    # read the identity view
    g, yraw, gs, gt, k1, k0 = gview_inputs(rng, (n, h, w, cout))
    bufs = [ctx.array(v) for v in (g, yraw, gs, gt, k1, k0)]
    gv = H.gview(*bufs, act=RELU6)
    ctx.call("ssdseg_gview_materialize", gv, cout, m, cout)
    dy = bufs[0].download()
    samp = sample_rows(rng, m, 2000)
    want = dy64(g.reshape(m, cout)[samp], yraw.reshape(m, cout)[samp], gs, gt, k1, k0)
    assert np.abs(dy.reshape(m, cout)[samp] - want).max() < 2e-5 * np.abs(want).max()
    gid = H.gview(bufs[0])
    ddx, ddw = ctx.empty((n, h, w, cin)), ctx.empty(wgt.shape)
    ctx.call("ssdseg_conv3x3_bwd_data", gid, dw_, ddx, cin, n, h, w, cin, cout, 0)
    dxg = ddx.download()
    wt = np.transpose(w64, (0, 1, 3, 2))          # [kh][kw][cout][cin]
    dx_s = conv_at(lambda nn, hh, ww: dy[nn, hh, ww].astype(np.float64), wt, -1)
    assert np.abs(dxg[pix[:, 0], pix[:, 1], pix[:, 2]] - dx_s).max() < 2e-5 * np.abs(dx_s).max()
    assert np.isfinite(dxg).all()
    del dxg
    ctx.call("ssdseg_conv3x3_bwd_weight", H.view(dx_, dsc, dsh, RELU6), cin, gid, ddw, n, h, w, cin, cout)
    dwg = ddw.download().astype(np.float64)
    assert np.isfinite(dwg).all()
    # per tap: dW[kh][kw] = sum over pixels a(shifted) ^T dy -- exact entries + projections over ALL pixels, image by image
    ent = [(int(rng.integers(0, cin)), int(rng.integers(0, cout))) for _ in range(6)] + [(0, 0), (cin - 1, cout - 1)]
    us = rng.normal(0, 1, (2, cin)); vs = rng.normal(0, 1, (2, cout))
    e_acc = np.zeros((3, 3, len(ent))); p_acc = np.zeros((3, 3, 2))
    for i in range(n):
        ai = np.zeros((h + 2, w + 2, cin))
        ai[1:h + 1, 1:w + 1] = act64(x[i], sc, sh)
        dyi = dy[i].astype(np.float64)
        au = ai @ us.T                         # (h+2, w+2, 2)
        dv = dyi @ vs.T                        # (h, w, 2)
        for kh in range(3):
            for kw in range(3):
                sub = ai[kh:kh + h, kw:kw + w]
                for j, (kk, nn) in enumerate(ent):
                    e_acc[kh, kw, j] += (sub[:, :, kk] * dyi[:, :, nn]).sum()
                p_acc[kh, kw] += (au[kh:kh + h, kw:kw + w] * dv).sum(axis=(0, 1))
    got_e = np.array([[[dwg[kh, kw, kk, nn] for kk, nn in ent] for kw in range(3)] for kh in range(3)])
    assert np.abs(got_e - e_acc).max() < 5e-5 * np.abs(dwg).max()
    got_p = np.einsum("ik,hwkn,in->hwi", us, dwg, vs)
    assert np.abs(got_p - p_acc).max() < 5e-5 * np.abs(us).sum(axis=1).max() * np.abs(dwg).max() * np.sqrt(cout)


# ------------------------------------------------------------------------------------------------ whole step, batch 32, 480x640
@pytest.mark.parametrize("workload", ["full", "shufflenet", "shufflenet-q1fixed"])
def test_full_train_step_batch32_480x640_properties(ctx, workload):
    """configs[2] (MobileNetV2) and configs[4]'s per-GPU share (ShuffleNetV2-1x) in both forms SURVEY.md 8(d) allows: reference quirk
    Q1 kept (the heads' ReLU(max 0) makes every class probability 0.25, so the hard-negative pool is ONE big tie -- the selection
    is decided purely by the lowest-index-first rule, on the device and in the oracle -- and every gradient is exactly zero), and
    Q1 fixed (ReLU6 in the head blocks: non-zero gradients through the same kernels)"""
    import bench
    from ssdseglib import _hip as H
    step = bench.STEPS[workload](ctx, 32, 0, None)
    eng = step.eng
    P = eng.P
    p0 = P["params"].download(); s0 = P["state"].download()

    def run():
        P["params"].upload(p0); P["state"].upload(s0)
        P["adam_m"].zero_(); P["adam_v"].zero_(); P["step"] = 0
        step()
        ctx.sync()
        return P["grads"].download(), P["params"].download(), P["state"].download(), eng.losses()

    g1, p1, s1, l1 = run()
    g2, p2, s2, l2 = run()
    if workload != "shufflenet":
        # round 3: the up-sampled ASPP output is written straight into the zero-bordered input copy of the decoder conv
        # (ssdseg_bilinear_fwd_padded + ssdseg_conv3x3_fwd_saved_from; x4 tile kernel for MobileNetV2, the general kernel for
        # ShuffleNetV2's x2); with the padding pass over all 304 channels instead (SSDSEG_CONV3_PADFUSE=0) the step is the same
        # bits -- and so it is with the detection branch on the main stream, in layer order (SSDSEG_DET_SIDE=0)
        from ssdseglib import _engine as E
        conv = next(op for op in eng.ops if isinstance(op, E.Conv3Op) and op.xsaved is not None)
        assert conv.saved_from == 256 and any(isinstance(op, E.BilinearOp) and op.padded_out is not None for op in eng.ops)
        trunk, det, mask, join_before = eng._schedule()
        assert len(det) > 50 and len(mask) > 20 and join_before, (len(trunk), len(det), len(mask))
        for var in ("SSDSEG_CONV3_PADFUSE", "SSDSEG_DET_SIDE"):
            os.environ[var] = "0"
            try:
                g3, p3, s3, l3 = run()
            finally:
                del os.environ[var]
            assert np.array_equal(g1, g3) and np.array_equal(p1, p3) and np.array_equal(s1, s3) and l1 == l3, var
    assert np.isfinite(g1).all() and np.isfinite(p1).all() and np.isfinite(s1).all()
    if workload == "shufflenet":
        # quirk Q1 to the letter: ReLU(max_value=0.0) in every head block zeroes their activations AND their derivative, so every
        # gradient of the model is exactly zero and Adam leaves the weights alone; only the moving statistics move
        assert not g1.any() and np.array_equal(p1, p0) and not np.array_equal(s1, s0)
    else:
        assert np.abs(g1).max() > 0 and not np.array_equal(p1, p0)
    assert all(np.isfinite(v) for v in l1.values()), l1
    # fixed-order reductions everywhere, also with the weight gradients on the side stream: two steps from the same state agree bit for bit
    assert np.array_equal(g1, g2) and np.array_equal(p1, p2) and np.array_equal(s1, s2) and l1 == l2
    # encoder output on the device == the oracle's on the same ground truth: matching (labels) exact, offsets to rounding
    det = step.det
    labels, offsets = det.y_labels.download(), det.y_boxes.download()
    gt, cnt = step.gt.download(), step.cnt.download()
    corners = step.anchors.download()
    for b in range(32):
        l_ref, o_ref, _ = O.encode_targets(corners, gt[b, :cnt[b]], 4, 0.525, bench.STDS)
        assert np.array_equal(labels[b], l_ref), f"image {b}: anchor matching differs"
        assert np.abs(offsets[b] - o_ref).max() < 1e-5
    assert labels[..., 1:].sum() > 32
    # hard-negative mining on the device's own probabilities: the selected set equals the oracle's top-k on the same tensor
    probs = det.probs.buf.download().reshape(32, 9600, 4)
    keep = ctx.empty(32 * 9600, np.uint8)
    conf, loc = ctx.empty(32), ctx.empty(32)
    ctx.call("ssdseg_det_loss", det.y_labels, det.probs.buf, det.y_boxes, det.boxes.buf, 32, 9600, 4, 1.0 / 32, conf, loc, None, None, keep)
    l_ref, _, keep_ref = O.confidence_loss(labels, probs)
    got_keep = keep.download()
    assert got_keep.sum() == keep_ref.sum()
    assert np.array_equal(got_keep, keep_ref), f"{(got_keep != keep_ref).sum()} mining decisions differ"
    assert np.abs(conf.download() - l_ref).max() < 1e-4 * np.abs(l_ref).max()
