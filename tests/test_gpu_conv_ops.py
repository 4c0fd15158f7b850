"""GPU parity: HIP conv / BN kernels (through the C-ABI) vs the NumPy oracle on the same seeded inputs.

Tolerance: north_star asks 1e-3 relative for conv/loss floats; single ops are held to 2e-5 of the tensor's
max magnitude (fp32 accumulation-order noise), reductions over millions of elements to 1e-4.
"""
import numpy as np
import pytest

from oracle import np_ops as O

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def make_view_inputs(rng, shape, act):
    """raw tensor + per-channel affine such that the activation clips a good share of the values"""
    c = shape[-1]
    x = rng.normal(0, 2.0, shape).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
    shift = rng.uniform(-1, 3, c).astype(np.float32)
    z = x.astype(np.float64) * scale + shift   # keep pre-activations off the ReLU6 thresholds (see make_gview_inputs)
    x = np.where((np.abs(z) < 1e-3) | (np.abs(z - 6) < 1e-3), x + np.float32(0.05), x).astype(np.float32)
    a = O.act_fwd(x * scale + shift, act)
    return x, scale, shift, a


def make_gview_inputs(rng, shape, act):
    """g, raw y, BN-backward coefficients -> dy exactly as the kernels must form it"""
    c = shape[-1]
    g = rng.normal(0, 1, shape).astype(np.float32)
    y = rng.normal(0, 2, shape).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
    shift = rng.uniform(-1, 3, c).astype(np.float32)
    # keep pre-activations away from the ReLU6 thresholds: the mask of an element within one rounding of 0 or 6 depends on
    # fma-vs-two-roundings, and with 1e7 elements a few such flips (each worth |g*scale|) would swamp the comparison
    z = y.astype(np.float64) * scale + shift
    y = np.where((np.abs(z) < 1e-3) | (np.abs(z - 6) < 1e-3), y + np.float32(0.05), y).astype(np.float32)
    k1 = rng.normal(0, 0.1, c).astype(np.float32)
    k0 = rng.normal(0, 0.1, c).astype(np.float32)
    dy = scale * O.act_mask(y * scale + shift, act) * g + k1 * y + k0
    return g, y, scale, shift, k1, k0, dy.astype(np.float32)


DW_CASES = [
    # n, h, w, c, stride, dilation
    (2, 15, 20, 32, 1, 1),
    (2, 15, 20, 32, 2, 1),     # 15 -> 8 pads (1,1); 20 -> 10 pads (0,1)
    (1, 30, 40, 144, 2, 1),    # 36 channel vectors
    (2, 8, 10, 960, 1, 1),     # 240 channel vectors
    (1, 9, 7, 8, 2, 1),        # odd x odd, pads (1,1)/(1,1)
    (1, 10, 7, 8, 2, 1),       # pads (0,1)/(1,1)
    (1, 9, 8, 8, 2, 1),        # pads (1,1)/(0,1)
    (2, 30, 40, 64, 1, 3),     # atrous
    (1, 30, 40, 32, 1, 12),
    (1, 30, 40, 576, 1, 6),    # the ASPP shape: 36 interleaved 5x7 sub-grids, three 192-channel chunks
    (2, 7, 5, 8, 1, 3),        # ragged sub-grids (3x2, 2x2, 2x1 ...)
    (1, 4, 5, 8, 1, 12),       # dilation > image: every sub-grid is one pixel, only the centre tap lands
    (1, 5, 6, 1284, 1, 1),     # > 256 channel vectors (grid.y = 2)
    (1, 1, 1, 4, 1, 1),
]


@pytest.mark.parametrize("n,h,w,c,s,d", DW_CASES)
@pytest.mark.parametrize("act", [O.ACT_RELU6, O.ACT_NONE])
def test_dwconv_fwd_bwd(ctx, rng, kernel_family, n, h, w, c, s, d, act):
    from ssdseglib import _hip as H
    x, sc, sh, a = make_view_inputs(rng, (n, h, w, c), act)
    wgt = rng.normal(0, 0.3, (3, 3, c)).astype(np.float32)
    y_ref = O.dwconv_fwd(a, wgt, s, d)
    ho, wo = y_ref.shape[1:3]
    dx_, dsc, dsh = ctx.array(x), ctx.array(sc), ctx.array(sh)
    dw_ = ctx.array(wgt)
    dy_ = ctx.empty(y_ref.shape)
    nparts = ctx.parts("ssdseg_dwconv_parts", n, h, w, c, s, d)
    stats = ctx.empty((nparts, 2, c))
    ctx.call("ssdseg_dwconv_fwd", H.view(dx_, dsc, dsh, act), dw_, dy_, n, h, w, c, s, d, stats)
    y = dy_.download()
    assert rel_err(y, y_ref) < 2e-5
    st = stats.download().astype(np.float64).sum(axis=0)
    assert rel_err(st[0], y_ref.sum(axis=(0, 1, 2), dtype=np.float64)) < 1e-4 or np.abs(st[0] - y_ref.sum(axis=(0, 1, 2))).max() < 1e-3
    assert rel_err(st[1], (y_ref.astype(np.float64) ** 2).sum(axis=(0, 1, 2))) < 1e-4

    # backward: dy formed from a gradient view
    g, yraw, gs, gt, k1, k0, dy = make_gview_inputs(rng, y_ref.shape, O.ACT_RELU6)
    dx_ref, dw_ref = O.dwconv_bwd(a, wgt, dy, s, d)
    bufs = [ctx.array(v) for v in (g, yraw, gs, gt, k1, k0)]
    gv = H.gview(*bufs, act=O.ACT_RELU6)
    ddx = ctx.empty(x.shape)
    ddw = ctx.empty(wgt.shape)
    ctx.call("ssdseg_dwconv_bwd", H.view(dx_, dsc, dsh, act), dw_, gv, ddx, ddw, n, h, w, c, s, d, 0)
    assert rel_err(ddx.download(), dx_ref) < 2e-5
    assert rel_err(ddw.download(), dw_ref) < 1e-4
    # accumulate into an existing gradient (fan-out taps)
    base = rng.normal(0, 1, x.shape).astype(np.float32)
    ddx.upload(base)
    ctx.call("ssdseg_dwconv_bwd", H.view(dx_, dsc, dsh, act), dw_, gv, ddx, ddw, n, h, w, c, s, d, 1)
    assert rel_err(ddx.download(), dx_ref + base) < 2e-5
    # identity gradient view
    gid = H.gview(bufs[0])
    ctx.call("ssdseg_dwconv_bwd", H.view(dx_, dsc, dsh, act), dw_, gid, ddx, ddw, n, h, w, c, s, d, 0)
    dx_ref2, dw_ref2 = O.dwconv_bwd(a, wgt, g, s, d)
    assert rel_err(ddx.download(), dx_ref2) < 2e-5
    assert rel_err(ddw.download(), dw_ref2) < 1e-4
    # fused: the same backward + the BatchNorm-backward reduction of the layer feeding this conv (x is that BN's raw input)
    mean = x.mean(axis=(0, 1, 2), dtype=np.float64).astype(np.float32)
    invstd = (1.0 / np.sqrt(x.var(axis=(0, 1, 2), dtype=np.float64) + 1e-3)).astype(np.float32)
    outs = [ctx.empty(c) for _ in range(4)]
    ctx.call("ssdseg_dwconv_bwd_bn", H.view(dx_, dsc, dsh, act), dw_, gv, ddx, ddw, n, h, w, c, s, d, 0, ctx.array(mean), ctx.array(invstd), *outs)
    assert rel_err(ddx.download(), dx_ref) < 2e-5
    assert rel_err(ddw.download(), dw_ref) < 1e-4
    z = x.astype(np.float64) * sc + sh
    mg = dx_ref.astype(np.float64) * O.act_mask(z, act)
    xhat = (x.astype(np.float64) - mean) * invstd
    dbeta, dgamma = mg.sum(axis=(0, 1, 2)), (mg * xhat).sum(axis=(0, 1, 2))
    cnt = float(n * h * w)
    tol = 1e-4 * max(np.abs(dgamma).max(), np.abs(dbeta).max(), 1e-6)
    assert np.abs(outs[0].download() - dgamma).max() < tol
    assert np.abs(outs[1].download() - dbeta).max() < tol
    k1_ref = -sc.astype(np.float64) * dgamma * invstd / cnt
    k0_ref = sc.astype(np.float64) * (dgamma * invstd * mean - dbeta) / cnt
    assert np.abs(outs[2].download() - k1_ref).max() < 1e-4 * max(np.abs(k1_ref).max(), 1e-9)
    assert np.abs(outs[3].download() - k0_ref).max() < 1e-4 * max(np.abs(k0_ref).max(), 1e-9)
    # last of several consumers: dx already holds the others' gradients; the BN sums are over the completed dx
    ddx.upload(base)
    ctx.call("ssdseg_dwconv_bwd_bn", H.view(dx_, dsc, dsh, act), dw_, gv, ddx, ddw, n, h, w, c, s, d, 1, ctx.array(mean), ctx.array(invstd), *outs)
    tot = dx_ref.astype(np.float64) + base
    assert rel_err(ddx.download(), tot) < 2e-5
    mg = tot * O.act_mask(z, act)
    dbeta, dgamma = mg.sum(axis=(0, 1, 2)), (mg * xhat).sum(axis=(0, 1, 2))
    tol = 1e-4 * max(np.abs(dgamma).max(), np.abs(dbeta).max(), 1e-6)
    assert np.abs(outs[0].download() - dgamma).max() < tol and np.abs(outs[1].download() - dbeta).max() < tol


PW_CASES = [
    # m, k, n
    (640, 16, 96),
    (1000, 96, 24),      # row tail, n < 32
    (300, 24, 144),      # n = 4.5 tiles
    (257, 144, 32),
    (2048, 576, 160),
    (130, 960, 320),
    (512, 1280, 256),
    (77, 360, 24),
    (4096, 32, 16),
    (100, 8, 4),
    (1500, 64, 384),     # backward-data with a long reduction and few row tiles: wide tile + split-K (+ residual, accumulate)
    (1500, 960, 160),    # forward the same way (BN statistics come from the split-K reduce kernel)
    (70001, 24, 144),    # fused dx+dW kernel: more row tiles than blocks, ragged last tile, 4.5 column chunks
    (66000, 32, 192),    # fused, 6 chunks (largest shape the fused kernel takes)
    (300, 200, 64),      # deep steps (two 32-deep sub-tiles per barrier pair): reduction of 6.25 sub-tiles, both directions
    (9600, 960, 160),    # the 15x20-stage project conv as it runs in the bench
    (70001, 96, 264),    # >= 65536 rows: the eight-wave tile GEMM (pw_tile.h), ragged last row tile, two 132-column tiles forward
    (66000, 256, 256),   # eight-wave tile GEMM, one 256-column tile forward / two 128-column tiles backward (the decoder sepconv shape)
    (300, 72, 40),       # tile GEMM with a reduction that is not a multiple of 32 (72 = 2 steps + 8) and a single ragged column tile
    (153600, 32, 192),   # weight gradient on pw_wgrad_kernel<1,4,1,1> at a large split count: the shape (with 70001 x 24 x 144) on which a
                         # 16-byte store with an SGPR soffset returned wrong lanes 12-15 of every 16 in round 2 (gfx950 store-data hazard)
    (66000, 176, 72),    # input gradient with 176 output columns at >= 65,536 rows: the 4 x 2 wave grid with three-tile waves (88 ragged columns per wave group)
]


@pytest.mark.parametrize("m,k,n", PW_CASES)
def test_pwconv_fwd_bwd(ctx, rng, kernel_family, m, k, n):
    from ssdseglib import _hip as H
    act = O.ACT_RELU6
    x, sc, sh, a = make_view_inputs(rng, (m, k), act)
    wgt = (rng.normal(0, 1, (k, n)) / np.sqrt(k)).astype(np.float32)
    y_ref = a.astype(np.float64) @ wgt.astype(np.float64)
    dx_, dsc, dsh, dw_ = ctx.array(x), ctx.array(sc), ctx.array(sh), ctx.array(wgt)
    dy_ = ctx.empty((m, n))
    nparts = ctx.parts("ssdseg_pwconv_parts", m, n)
    stats = ctx.empty((nparts, 2, n))
    ctx.call("ssdseg_pwconv_fwd", H.view(dx_, dsc, dsh, act), k, dw_, dy_, n, m, k, n, stats)
    y = dy_.download()
    assert rel_err(y, y_ref) < 2e-5
    st = stats.download().astype(np.float64).sum(axis=0)
    assert np.abs(st[0] - y_ref.sum(axis=0)).max() < 1e-4 * max(1.0, np.abs(y_ref).sum(axis=0).max())
    assert rel_err(st[1], (y_ref ** 2).sum(axis=0)) < 1e-4
    # the engine's form: weights transposed for a whole table of layers in one launch, then handed to the forward -- bit-identical
    # (this case's matrix plus a second, ragged one in the same table; the copy is ignored where another kernel family runs)
    w2 = rng.normal(0, 1, (37, 50)).astype(np.float32)
    dw2, dwt, dwt2 = ctx.array(w2), ctx.zeros((n, k)), ctx.zeros((50, 37))
    table = ctx.array(np.array([[dw_.ptr, dwt.ptr, k, n], [dw2.ptr, dwt2.ptr, 37, 50]], dtype=np.int64))
    ctx.call("ssdseg_transpose_batch", table, 2, max(-(-k // 32) * -(-n // 32), 4), k * n + 37 * 50)
    assert np.array_equal(dwt.download(), wgt.T) and np.array_equal(dwt2.download(), w2.T)
    dy2, stats2 = ctx.empty((m, n)), ctx.empty((nparts, 2, n))
    ctx.call("ssdseg_pwconv_fwd_wt", H.view(dx_, dsc, dsh, act), k, dw_, dwt, dy2, n, m, k, n, stats2)
    assert np.array_equal(dy2.download(), y) and np.array_equal(stats2.download(), stats.download())
    # identity view, no stats
    ctx.call("ssdseg_pwconv_fwd", H.view(dx_), k, dw_, dy_, n, m, k, n, None)
    assert rel_err(dy_.download(), x.astype(np.float64) @ wgt.astype(np.float64)) < 2e-5

    g, yraw, gs, gt, k1, k0, dy = make_gview_inputs(rng, (m, n), O.ACT_RELU6)
    bufs = [ctx.array(v) for v in (g, yraw, gs, gt, k1, k0)]
    gv = H.gview(*bufs, act=O.ACT_RELU6)
    dxg = ctx.empty((m, k))
    ctx.call("ssdseg_pwconv_bwd_data", gv, n, dw_, dxg, k, m, k, n, None, 0, 0)
    dx_ref = dy.astype(np.float64) @ wgt.astype(np.float64).T
    assert rel_err(dxg.download(), dx_ref) < 2e-5
    res = rng.normal(0, 1, (m, k)).astype(np.float32)
    dres = ctx.array(res)
    base = rng.normal(0, 1, (m, k)).astype(np.float32)
    dxg.upload(base)
    ctx.call("ssdseg_pwconv_bwd_data", gv, n, dw_, dxg, k, m, k, n, dres, k, 1)
    assert rel_err(dxg.download(), dx_ref + res + base) < 2e-5

    dwg = ctx.empty((k, n))
    ctx.call("ssdseg_pwconv_bwd_weight", H.view(dx_, dsc, dsh, act), k, gv, n, dwg, m, k, n)
    dw_ref = a.astype(np.float64).T @ dy.astype(np.float64)
    assert rel_err(dwg.download(), dw_ref) < 5e-5
    # dx + dW in one call (one fused kernel for k <= 32, n <= 192), incl. residual + accumulate and the identity views.  Its
    # gradient view carries the conv's OWN forward output (the contract of ssdseg_pwconv_bwd: the fused kernel recomputes y from
    # in and w instead of reading it): y of the forward call above.  The incoming gradient is zeroed where the pre-activation
    # sits within 1e-4 of a ReLU6 threshold, so that a last-bit difference between a stored and a recomputed y cannot flip a
    # mask that matters.
    z_own = y.astype(np.float64) * gs + gt
    g_own = np.where((np.abs(z_own) < 1e-4) | (np.abs(z_own - 6) < 1e-4), np.float32(0), g).astype(np.float32)
    dy_own = gs * O.act_mask(z_own, act) * g_own.astype(np.float64) + k1 * y.astype(np.float64) + k0
    gv_own = H.gview(ctx.array(g_own), ctx.array(y), *bufs[2:], act=O.ACT_RELU6)
    dx_own, dw_own = dy_own @ wgt.astype(np.float64).T, a.astype(np.float64).T @ dy_own
    dwg.upload(np.zeros((k, n), np.float32))
    dxg.upload(base)
    ctx.call("ssdseg_pwconv_bwd", H.view(dx_, dsc, dsh, act), k, gv_own, n, dw_, dxg, k, dwg, m, k, n, dres, k, 1)
    assert rel_err(dxg.download(), dx_own + res + base) < 2e-5
    assert rel_err(dwg.download(), dw_own) < 5e-5
    ctx.call("ssdseg_pwconv_bwd", H.view(dx_), k, H.gview(bufs[0]), n, dw_, dxg, k, dwg, m, k, n, None, 0, 0)
    assert rel_err(dxg.download(), g.astype(np.float64) @ wgt.astype(np.float64).T) < 2e-5
    assert rel_err(dwg.download(), x.astype(np.float64).T @ g.astype(np.float64)) < 5e-5
    # dx + dW + the BatchNorm backward of the layer feeding the conv (x is that BN's raw input), fused in the float4 epilogue
    mean = x.mean(axis=0, dtype=np.float64).astype(np.float32)
    invstd = (1.0 / np.sqrt(x.var(axis=0, dtype=np.float64) + 1e-3)).astype(np.float32)
    outs = [ctx.empty(k) for _ in range(4)]
    ctx.call("ssdseg_pwconv_bwd_bn", H.view(dx_, dsc, dsh, act), k, gv, n, dw_, dxg, k, dwg, m, k, n, ctx.array(mean), ctx.array(invstd), *outs)
    assert rel_err(dxg.download(), dx_ref) < 2e-5
    assert rel_err(dwg.download(), dw_ref) < 5e-5
    z = x.astype(np.float64) * sc + sh
    mg = dx_ref * O.act_mask(z, act)
    xhat = (x.astype(np.float64) - mean) * invstd
    dbeta, dgamma = mg.sum(axis=0), (mg * xhat).sum(axis=0)
    tol = 1e-4 * max(np.abs(dgamma).max(), np.abs(dbeta).max(), 1e-6)
    assert np.abs(outs[0].download() - dgamma).max() < tol
    assert np.abs(outs[1].download() - dbeta).max() < tol
    k1_ref = -sc.astype(np.float64) * dgamma * invstd / m
    k0_ref = sc.astype(np.float64) * (dgamma * invstd * mean - dbeta) / m
    assert np.abs(outs[2].download() - k1_ref).max() < 1e-4 * max(np.abs(k1_ref).max(), 1e-9)
    assert np.abs(outs[3].download() - k0_ref).max() < 1e-4 * max(np.abs(k0_ref).max(), 1e-9)


def test_pwconv_strided_concat_slice(ctx, rng):
    """ldx/ldy: read a channel slice of a wider buffer and write into a slice of a concat buffer (K10)."""
    from ssdseglib import _hip as H
    m, k, n, ldx, ldy = 200, 48, 64, 80, 304
    xw = rng.normal(0, 1, (m, ldx)).astype(np.float32)
    wgt = rng.normal(0, 0.2, (k, n)).astype(np.float32)
    big = ctx.zeros((m, ldy))
    dxw = ctx.array(xw)
    ctx.call("ssdseg_pwconv_fwd", H.view(dxw.view(16, (m, k))), ldx, ctx.array(wgt), big.view(240, (m, n)), ldy, m, k, n, None)
    out = big.download()
    assert rel_err(out[:, 240:304], xw[:, 16:64].astype(np.float64) @ wgt) < 2e-5
    assert np.all(out[:, :240] == 0)


@pytest.mark.parametrize("m,c,parts", [(5000, 96, 700), (64, 24, 3), (100000, 16, 19200 // 8)])
def test_bn_finalize_apply_bwd(ctx, rng, m, c, parts):
    from ssdseglib import _hip as H
    y = rng.normal(0.5, 2.0, (m, c)).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, c).astype(np.float32)
    beta = rng.normal(0, 0.5, c).astype(np.float32)
    mm0 = rng.normal(0, 1, c).astype(np.float32)
    mv0 = rng.uniform(0.5, 2, c).astype(np.float32)
    z_ref, cache = O.bn_train_fwd(y, gamma, beta)
    mm_ref, mv_ref = O.bn_moving_update(mm0, mv0, cache)
    # channel stats kernel -> partials; also synthesise a many-rows partial table to hit the fold path
    dy_ = ctx.array(y)
    np_ = ctx.parts("ssdseg_channel_stats_parts", m, c)
    st = ctx.empty((np_, 2, c))
    ctx.call("ssdseg_channel_stats", dy_, c, m, c, st)
    tot = st.download().astype(np.float64).sum(axis=0)
    assert rel_err(tot[0], y.sum(axis=0, dtype=np.float64)) < 1e-5
    # spread the true sums over `parts` rows
    idx = np.arange(m) % parts
    tab = np.zeros((parts, 2, c), np.float64)
    np.add.at(tab[:, 0], idx, y.astype(np.float64))
    np.add.at(tab[:, 1], idx, y.astype(np.float64) ** 2)
    dtab = ctx.array(tab.astype(np.float32))
    bufs = {k: ctx.array(v) for k, v in dict(gamma=gamma, beta=beta, mm=mm0, mv=mv0).items()}
    mean, invstd, scale, shift = (ctx.empty(c) for _ in range(4))
    ctx.call("ssdseg_bn_finalize", dtab, parts, c, float(m), bufs["gamma"], bufs["beta"], 1e-3, 0.99, bufs["mm"], bufs["mv"],
             mean, invstd, scale, shift, 1)
    assert rel_err(mean.download(), cache["mean"]) < 1e-5
    assert rel_err(invstd.download(), cache["invstd"]) < 1e-5
    assert rel_err(scale.download(), cache["scale"]) < 1e-5
    assert np.abs(shift.download() - cache["shift"]).max() < 1e-5
    assert rel_err(bufs["mm"].download(), mm_ref) < 1e-5
    assert rel_err(bufs["mv"].download(), mv_ref) < 1e-5
    # inference affine from the moving statistics
    s2, t2 = ctx.empty(c), ctx.empty(c)
    ctx.call("ssdseg_bn_finalize", None, 0, c, 0.0, bufs["gamma"], bufs["beta"], 1e-3, 0.99, bufs["mm"], bufs["mv"], None, None, s2, t2, 0)
    s_ref, t_ref = O.bn_infer_affine(gamma, beta, bufs["mm"].download(), bufs["mv"].download())
    assert rel_err(s2.download(), s_ref) < 1e-5 and np.abs(t2.download() - t_ref).max() < 1e-5

    # apply (+ residual)
    res = rng.normal(0, 1, (m, c)).astype(np.float32)
    out = ctx.empty((m, c))
    ctx.call("ssdseg_bn_apply", H.view(dy_, scale, shift, O.ACT_RELU6), c, H.view(ctx.array(res)), c, out, c, m, c)
    assert np.abs(out.download() - (O.act_fwd(z_ref, O.ACT_RELU6) + res)).max() < 2e-5

    # backward: reduce -> coefficients -> dy through the gradient view formula
    g = rng.normal(0, 1, (m, c)).astype(np.float32)
    dz = g * O.act_mask(z_ref, O.ACT_RELU6)
    dy_ref, dgamma_ref, dbeta_ref = O.bn_train_bwd(dz, y, gamma, cache)
    dgamma, dbeta, k1, k0 = (ctx.empty(c) for _ in range(4))
    ctx.call("ssdseg_bn_bwd_reduce", ctx.array(g), c, dy_, c, m, c, scale, shift, mean, invstd, O.ACT_RELU6, dgamma, dbeta, k1, k0)
    assert rel_err(dgamma.download(), dgamma_ref) < 1e-4
    assert rel_err(dbeta.download(), dbeta_ref) < 1e-4
    sc = scale.download(); sh = shift.download()
    dy = sc * O.act_mask(y * sc + sh, O.ACT_RELU6) * g + k1.download() * y + k0.download()
    assert np.abs(dy - dy_ref).max() < 1e-4 * max(1.0, np.abs(dy_ref).max())


@pytest.mark.parametrize("form", ["direct", "gemm"])      # csrc/stem.hip (default for <= 64 output channels) / the implicit GEMM of csrc/gemm.hip
@pytest.mark.parametrize("n,h,w,cout,bias", [(2, 48, 64, 32, False), (1, 15, 21, 24, True), (3, 480, 640, 32, False), (2, 33, 47, 40, False),
                                             (9, 480, 64, 32, False)])      # 2,160 output rows: more than the direct kernel's 2,048 blocks
def test_stem_conv(ctx, rng, monkeypatch, n, h, w, cout, bias, form):
    from ssdseglib import _hip as H
    monkeypatch.setenv("SSDSEG_STEM_DIRECT", "1" if form == "direct" else "0")
    x = rng.integers(0, 256, (n, h, w, 3)).astype(np.float32)
    wgt = rng.normal(0, 0.3, (3, 3, 3, cout)).astype(np.float32)
    b = rng.normal(0, 0.3, cout).astype(np.float32) if bias else None
    xr = O.rescale(x)
    y_ref = O.conv2d_fwd(xr, wgt, 2, 1, b)
    dx_, dw_ = ctx.array(x), ctx.array(wgt)
    db_ = ctx.array(b) if bias else None
    y = ctx.empty(y_ref.shape)
    nparts = ctx.parts("ssdseg_stem_conv_parts", n, h, w, cout)
    stats = ctx.empty((nparts, 2, cout))
    # BN statistics are only requested for the bias-free stem (a biased conv is never followed by BatchNormalization)
    ctx.call("ssdseg_stem_conv_fwd", dx_, dw_, db_, y, n, h, w, 3, cout, 1.0 / 127.5, -1.0, None if bias else stats)
    assert rel_err(y.download(), y_ref) < 2e-5
    if not bias:
        st = stats.download().astype(np.float64).sum(axis=0)
        assert rel_err(st[1], (y_ref.astype(np.float64) ** 2).sum(axis=(0, 1, 2))) < 1e-4
    g, yraw, gs, gt, k1, k0, dy = make_gview_inputs(rng, y_ref.shape, O.ACT_RELU6)
    bufs = [ctx.array(v) for v in (g, yraw, gs, gt, k1, k0)]
    dwg = ctx.empty(wgt.shape)
    ctx.call("ssdseg_stem_conv_bwd_weight", dx_, H.gview(*bufs, act=O.ACT_RELU6), dwg, None, n, h, w, 3, cout, 1.0 / 127.5, -1.0)
    _, dw_ref, _ = O.conv2d_bwd(xr.astype(np.float64), wgt.astype(np.float64), dy.astype(np.float64), 2, 1)
    assert rel_err(dwg.download(), dw_ref) < 1e-4
    # biased stem (ShuffleNetV2, reference models.py:628): no BatchNormalization follows, so the gradient view is the identity
    dbg = ctx.empty(cout)
    ctx.call("ssdseg_stem_conv_bwd_weight", dx_, H.gview(bufs[0]), dwg, dbg, n, h, w, 3, cout, 1.0 / 127.5, -1.0)
    _, dw_ref, db_ref = O.conv2d_bwd(xr.astype(np.float64), wgt.astype(np.float64), g.astype(np.float64), 2, 1)
    assert rel_err(dwg.download(), dw_ref) < 1e-4
    assert rel_err(dbg.download(), db_ref) < 1e-4
