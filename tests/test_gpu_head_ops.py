"""GPU parity of the head / loss / anchor kernels (through the C-ABI) vs the NumPy oracle.

Float kernels: tolerance stated per assert (north_star: 1e-3 rel).  Index kernels (encode matching, top-k mining
mask, NMS selection, segmentation suppression): bit-exact.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import np_ops as O

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def view_inputs(rng, shape, act):
    c = shape[-1]
    x = rng.normal(0, 2.0, shape).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
    shift = rng.uniform(-1, 3, c).astype(np.float32)
    return x, scale, shift, O.act_fwd(x * scale + shift, act)


def gview_inputs(rng, shape, act):
    c = shape[-1]
    g = rng.normal(0, 1, shape).astype(np.float32)
    y = rng.normal(0, 2, shape).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
    shift = rng.uniform(-1, 3, c).astype(np.float32)
    k1 = rng.normal(0, 0.1, c).astype(np.float32)
    k0 = rng.normal(0, 0.1, c).astype(np.float32)
    dy = scale * O.act_mask(y * scale + shift, act) * g + k1 * y + k0
    return (g, y, scale, shift, k1, k0), dy.astype(np.float32)


def test_conv3x3_wino4_plain_store_path(ctx, rng, monkeypatch):
    """Outputs of 2 GiB and more leave the F(4x4, 3x3) kernel through 64-bit addresses instead of range-checked buffer stores
    (conv3_wino4.h, `accumulate == 2`); SSDSEG_W4_PLAIN_STORES=1 takes that path at a size the oracle finishes in seconds."""
    from ssdseglib import _hip as H
    monkeypatch.setenv("SSDSEG_CONV3_NARROW", "0")
    monkeypatch.setenv("SSDSEG_CONV3_TILE", "1")
    monkeypatch.setenv("SSDSEG_CONV3_WINOGRAD", "1")
    monkeypatch.setenv("SSDSEG_CONV3_F4", "1")
    monkeypatch.setenv("SSDSEG_W4_PLAIN_STORES", "1")
    n, h, w, cin, cout = 2, 13, 18, 48, 80          # partial tiles on both edges, a partial 32-channel tile either way (reductions of 48 / 80: whole 16-channel steps)
    act = O.ACT_RELU6
    x, sc, sh, a = view_inputs(rng, (n, h, w, cin), act)
    wgt = (rng.normal(0, 1, (3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    y_ref = O.conv2d_fwd(a.astype(np.float64), wgt.astype(np.float64))
    dx_, dsc, dsh, dw_ = ctx.array(x), ctx.array(sc), ctx.array(sh), ctx.array(wgt)
    y = ctx.empty(y_ref.shape)
    nparts = ctx.parts("ssdseg_conv3x3_parts", n, h, w, cin, cout)
    stats = ctx.empty((nparts, 2, cout))
    ctx.call("ssdseg_conv3x3_fwd", H.view(dx_, dsc, dsh, act), cin, dw_, y, n, h, w, cin, cout, stats)
    assert rel_err(y.download(), y_ref) < 5e-5
    st = stats.download().astype(np.float64).sum(axis=0)
    assert rel_err(st[1], (y_ref ** 2).sum(axis=(0, 1, 2))) < 1e-4
    dy = rng.normal(0, 1, y_ref.shape).astype(np.float32)
    dx_ref, _, _ = O.conv2d_bwd(a.astype(np.float64), wgt.astype(np.float64), dy.astype(np.float64))
    ddx = ctx.empty(x.shape)
    ctx.call("ssdseg_conv3x3_bwd_data", H.gview(ctx.array(dy)), dw_, ddx, cin, n, h, w, cin, cout, 0)
    assert rel_err(ddx.download(), dx_ref) < 5e-5


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 9, 11, 16, 32), (1, 12, 16, 304, 256), (2, 8, 8, 256, 4), (1, 5, 5, 24, 8), (1, 1, 1, 8, 8),
                                            (18, 120, 22, 256, 4),  # 2,160 image rows: the streaming logits-conv gradient (conv3n.hip) walks > 2,048 rows per launch, 22 = five 4-pixel trips + 2
                                            (3, 14, 40, 72, 96),    # 40 = one full + one partial 32-pixel step / tile per image row
                                            (2, 17, 33, 40, 200),   # odd number of 8-channel steps; two 100-column tiles of a 128-wide block
                                            (1, 8, 32, 12, 16),     # cin not a multiple of 8: the implicit-GEMM forward, halo-tile backward
                                            (2, 6, 64, 80, 72),     # even h, w a multiple of 32: the Winograd weight gradient ("wino"), partial 64-channel patches, 3 splits
                                            (3, 2, 32, 64, 64),     # one tile row per image: every step crosses an image border of the padded copy
                                            (5, 4, 32, 132, 72),    # Winograd weight gradient: 6 patches x 10 steps = 2 full chunks of 4 + a 2-step tail dealt to 3 blocks, each working on two patches in turn
                                            (9, 2, 32, 72, 136)])   # ... 9 steps: a 1-step tail, one tail block walks four patches, the other two
# kernel family: "narrow" = cout <= 8 as tap-expanded pointwise GEMMs (default for those shapes); "tile" = the halo-tile kernels
# (conv3_tile.h); "wino" = the Winograd F(2x2, 3x3) kernels forced at every size (conv3_wino.h, default for the large layers: same
# tolerance -- its transforms are additions and halvings); "gemm" = the implicit-GEMM kernels
# "wino4" = forward / input gradient as Winograd F(4x4, 3x3) (conv3_wino4.h) wherever a tile geometry exists: constants up to 8 in
# the transforms, 1.2e-5 of the output scale at 304 channels (scripts/study/winograd_f4x4_error.py) -- held to 5e-5
@pytest.mark.parametrize("family", ["narrow", "tile", "wino", "wino4", "gemm"])
def test_conv3x3_fwd_bwd(ctx, rng, monkeypatch, n, h, w, cin, cout, family):
    from ssdseglib import _hip as H
    if family == "narrow" and cout > 8:
        pytest.skip("tap-expanded form only for cout <= 8")
    monkeypatch.setenv("SSDSEG_CONV3_NARROW", "1" if family == "narrow" else "0")
    monkeypatch.setenv("SSDSEG_CONV3_TILE", "0" if family == "gemm" else "1")
    monkeypatch.setenv("SSDSEG_CONV3_WINOGRAD", "1" if family in ("wino", "wino4") else "0")
    monkeypatch.setenv("SSDSEG_CONV3_F4", "1" if family == "wino4" else "0")
    ctol = 5e-5 if family == "wino4" else 2e-5
    act = O.ACT_RELU6
    x, sc, sh, a = view_inputs(rng, (n, h, w, cin), act)
    wgt = (rng.normal(0, 1, (3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    y_ref = O.conv2d_fwd(a.astype(np.float64), wgt.astype(np.float64))
    dx_, dsc, dsh, dw_ = ctx.array(x), ctx.array(sc), ctx.array(sh), ctx.array(wgt)
    y = ctx.empty(y_ref.shape)
    nparts = ctx.parts("ssdseg_conv3x3_parts", n, h, w, cin, cout)
    stats = ctx.empty((nparts, 2, cout))
    ctx.call("ssdseg_conv3x3_fwd", H.view(dx_, dsc, dsh, act), cin, dw_, y, n, h, w, cin, cout, stats)
    assert rel_err(y.download(), y_ref) < ctol
    st = stats.download().astype(np.float64).sum(axis=0)
    assert rel_err(st[1], (y_ref ** 2).sum(axis=(0, 1, 2))) < 1e-4
    assert np.abs(st[0] - y_ref.sum(axis=(0, 1, 2))).max() < 1e-4 * np.abs(y_ref).sum(axis=(0, 1, 2)).max()
    # forward from a channel slice of a wider input buffer (the decoder conv reads the 304-channel concat)
    ldi = cin + 8
    xw = np.zeros((n, h, w, ldi), np.float32)
    xw[..., 4:4 + cin] = x
    dxw = ctx.array(xw)
    ctx.call("ssdseg_conv3x3_fwd", H.view(dxw.view(4, (dxw.size - 4,)), dsc, dsh, act), ldi, dw_, y, n, h, w, cin, cout, None)
    assert rel_err(y.download(), y_ref) < ctol
    gv, dy = gview_inputs(rng, y_ref.shape, O.ACT_RELU6)
    bufs = [ctx.array(v) for v in gv]
    dx_ref, dw_ref, _ = O.conv2d_bwd(a.astype(np.float64), wgt.astype(np.float64), dy.astype(np.float64))
    ddx = ctx.empty(x.shape)
    ctx.call("ssdseg_conv3x3_bwd_data", H.gview(*bufs, act=O.ACT_RELU6), dw_, ddx, cin, n, h, w, cin, cout, 0)
    assert rel_err(ddx.download(), dx_ref) < ctol
    base = rng.normal(0, 1, x.shape).astype(np.float32)
    ddx.upload(base)
    ctx.call("ssdseg_conv3x3_bwd_data", H.gview(*bufs, act=O.ACT_RELU6), dw_, ddx, cin, n, h, w, cin, cout, 1)
    assert rel_err(ddx.download(), dx_ref + base) < ctol
    # input gradient + the BatchNorm backward of the layer feeding the conv (x is that BN's raw input): in the GEMM epilogue for the
    # narrow form, conv + separate reduction otherwise -- the same numbers either way
    mean = x.mean(axis=(0, 1, 2), dtype=np.float64).astype(np.float32)
    invstd = (1.0 / np.sqrt(x.var(axis=(0, 1, 2), dtype=np.float64) + 1e-3)).astype(np.float32)
    outs = [ctx.empty(cin) for _ in range(4)]
    ddx.upload(base)
    ctx.call("ssdseg_conv3x3_bwd_data_bn", H.view(dx_, dsc, dsh, act), H.gview(*bufs, act=O.ACT_RELU6), dw_, ddx, cin, n, h, w, cin, cout,
             ctx.array(mean), ctx.array(invstd), *outs)
    assert rel_err(ddx.download(), dx_ref) < ctol
    cnt = n * h * w
    mg = dx_ref * O.act_mask(x.astype(np.float64) * sc + sh, act)
    xhat = (x.astype(np.float64) - mean) * invstd
    dbeta, dgamma = mg.sum(axis=(0, 1, 2)), (mg * xhat).sum(axis=(0, 1, 2))
    tol = 1e-4 * max(np.abs(dgamma).max(), np.abs(dbeta).max(), 1e-6)
    assert np.abs(outs[0].download() - dgamma).max() < tol and np.abs(outs[1].download() - dbeta).max() < tol
    k1_ref = -sc.astype(np.float64) * dgamma * invstd / cnt
    k0_ref = sc.astype(np.float64) * (dgamma * invstd * mean - dbeta) / cnt
    assert np.abs(outs[2].download() - k1_ref).max() < 1e-4 * max(np.abs(k1_ref).max(), 1e-9)
    assert np.abs(outs[3].download() - k0_ref).max() < 1e-4 * max(np.abs(k0_ref).max(), 1e-9)
    if family == "narrow" and cin == 256 and cout == 4:
        # that was the streaming kernel of csrc/conv3n.hip (default for 256 -> 4); the tap-expanded GEMM form of the same entry point:
        monkeypatch.setenv("SSDSEG_CONV3N_DIRECT", "0")
        outs2 = [ctx.empty(cin) for _ in range(4)]
        ddx.upload(base)
        ctx.call("ssdseg_conv3x3_bwd_data_bn", H.view(dx_, dsc, dsh, act), H.gview(*bufs, act=O.ACT_RELU6), dw_, ddx, cin, n, h, w, cin, cout,
                 ctx.array(mean), ctx.array(invstd), *outs2)
        assert rel_err(ddx.download(), dx_ref) < ctol
        assert np.abs(outs2[0].download() - dgamma).max() < tol and np.abs(outs2[1].download() - dbeta).max() < tol
        monkeypatch.delenv("SSDSEG_CONV3N_DIRECT")
    # the engine's form: the BatchNorm gradient view materialised once, then the identity view (the halo-tile kernel's input),
    # written into a channel slice of a wider (concat) gradient buffer, overwrite and accumulate
    dmat = ctx.array(dy)
    ldx = cin + 8
    wide = ctx.zeros((n * h * w, ldx))
    ctx.call("ssdseg_conv3x3_bwd_data", H.gview(dmat), dw_, wide.view(4, (wide.size - 4,)), ldx, n, h, w, cin, cout, 0)
    got = wide.download().reshape(n, h, w, ldx)
    assert rel_err(got[..., 4:4 + cin], dx_ref) < ctol and np.all(got[..., :4] == 0) and np.all(got[..., 4 + cin:] == 0)
    ctx.call("ssdseg_conv3x3_bwd_data", H.gview(dmat), dw_, wide.view(4, (wide.size - 4,)), ldx, n, h, w, cin, cout, 1)
    assert rel_err(wide.download().reshape(n, h, w, ldx)[..., 4:4 + cin], 2 * dx_ref) < ctol
    ddw = ctx.empty(wgt.shape)
    # identity gradient view (the engine's form): the halo-tile weight-gradient kernel in the "tile" family; input read from a
    # channel slice of the wider buffer
    ctx.call("ssdseg_conv3x3_bwd_weight", H.view(dxw.view(4, (dxw.size - 4,)), dsc, dsh, act), ldi, H.gview(dmat), ddw, n, h, w, cin, cout)
    assert rel_err(ddw.download(), dw_ref) < 5e-5
    ddw.upload(np.zeros(wgt.shape, np.float32))
    ctx.call("ssdseg_conv3x3_bwd_weight", H.view(dx_, dsc, dsh, act), cin, H.gview(*bufs, act=O.ACT_RELU6), ddw, n, h, w, cin, cout)
    assert rel_err(ddw.download(), dw_ref) < 5e-5      # BatchNorm gradient view: all nine taps in one pass (conv3_wgrad.h; twelve waves for cout > 32)
    monkeypatch.setenv("SSDSEG_CONV3_WGRAD", "nine")   # one wave per tap for every width
    ddw.upload(np.zeros(wgt.shape, np.float32))
    ctx.call("ssdseg_conv3x3_bwd_weight", H.view(dx_, dsc, dsh, act), cin, H.gview(*bufs, act=O.ACT_RELU6), ddw, n, h, w, cin, cout)
    assert rel_err(ddw.download(), dw_ref) < 5e-5
    monkeypatch.setenv("SSDSEG_CONV3_WGRAD", "taps")   # the nine shifted weight-gradient GEMMs
    ddw.upload(np.zeros(wgt.shape, np.float32))
    ctx.call("ssdseg_conv3x3_bwd_weight", H.view(dx_, dsc, dsh, act), cin, H.gview(*bufs, act=O.ACT_RELU6), ddw, n, h, w, cin, cout)
    assert rel_err(ddw.download(), dw_ref) < 5e-5


def test_gap(ctx, rng):
    from ssdseglib import _hip as H
    n, h, w, c = 3, 30, 40, 576
    x, sc, sh, a = view_inputs(rng, (n, h, w, c), O.ACT_RELU6)
    out = ctx.empty((n, c))
    ctx.call("ssdseg_gap_fwd", H.view(ctx.array(x), ctx.array(sc), ctx.array(sh), O.ACT_RELU6), out, n, h * w, c)
    assert rel_err(out.download(), O.gap_fwd(a)[:, 0, 0]) < 1e-5
    g = rng.normal(0, 1, (n, 1, 1, c)).astype(np.float32)
    base = rng.normal(0, 1, (n, h, w, c)).astype(np.float32)
    dx = ctx.array(base)
    ctx.call("ssdseg_gap_bwd", ctx.array(g), dx, n, h * w, c, 1)
    assert rel_err(dx.download(), O.gap_bwd(g, h, w) + base) < 1e-6
    ctx.call("ssdseg_gap_bwd", ctx.array(g), dx, n, h * w, c, 0)
    assert rel_err(dx.download(), O.gap_bwd(g, h, w)) < 1e-6


@pytest.mark.parametrize("n,h,w,c,fy,fx", [(2, 6, 8, 16, 4, 4), (2, 1, 1, 256, 30, 40), (1, 5, 7, 8, 2, 8), (1, 3, 3, 4, 1, 1), (3, 2, 1, 72, 4, 4),
                                             (2, 5, 7, 8, 4, 4), (1, 30, 40, 16, 4, 4)])
def test_bilinear(ctx, rng, monkeypatch, n, h, w, c, fy, fx):
    from ssdseglib import _hip as H
    if fy == 4 and fx == 4:
        # the x4 forward kernel (one input pixel's 4x4 outputs per thread, 3x3 inputs loaded once) == the general gather kernel, bit for bit
        xx, s1, s2, _ = view_inputs(rng, (n, h, w, c), O.ACT_RELU6)
        bufs = [ctx.array(v) for v in (xx, s1, s2)]
        o4, og = ctx.empty((n, h * 4, w * 4, c)), ctx.empty((n, h * 4, w * 4, c))
        ctx.call("ssdseg_bilinear_fwd", H.view(*bufs, O.ACT_RELU6), c, o4, c, n, h, w, c, 4, 4)
        monkeypatch.setenv("SSDSEG_BILINEAR", "gather")
        ctx.call("ssdseg_bilinear_fwd", H.view(*bufs, O.ACT_RELU6), c, og, c, n, h, w, c, 4, 4)
        monkeypatch.delenv("SSDSEG_BILINEAR")
        np.testing.assert_array_equal(o4.download(), og.download())
    x, sc, sh, a = view_inputs(rng, (n, h, w, c), O.ACT_RELU6)
    ref = O.bilinear_fwd(a, fy, fx)
    ldo = c + 8                                     # write into a slice of a wider (concat) buffer
    out = ctx.zeros((n * h * fy * w * fx, ldo))
    ctx.call("ssdseg_bilinear_fwd", H.view(ctx.array(x), ctx.array(sc), ctx.array(sh), O.ACT_RELU6), c, out.view(4, (out.size - 4,)), ldo,
             n, h, w, c, fy, fx)
    got = out.download().reshape(n, h * fy, w * fx, ldo)
    assert np.abs(got[..., 4:4 + c] - ref).max() < 1e-5
    assert np.all(got[..., :4] == 0) and np.all(got[..., 4 + c:] == 0)
    g = rng.normal(0, 1, ref.shape).astype(np.float32)
    dx_ref = O.bilinear_bwd(g.astype(np.float64), fy, fx)
    dx = ctx.empty((n, h, w, c))
    ctx.call("ssdseg_bilinear_bwd", ctx.array(g), c, dx, c, n, h, w, c, fy, fx, 0)
    assert rel_err(dx.download(), dx_ref) < 1e-5
    if fy == 4 and fx == 4:
        # the x4 backward kernel (a 2x2 block of input pixels per thread, the 12x12 union window loaded once; odd sizes: partial
        # blocks) against the general gather kernel: same terms, another summation order
        dg = ctx.empty((n, h, w, c))
        monkeypatch.setenv("SSDSEG_BILINEAR", "gather")
        ctx.call("ssdseg_bilinear_bwd", ctx.array(g), c, dg, c, n, h, w, c, fy, fx, 0)
        monkeypatch.delenv("SSDSEG_BILINEAR")
        assert rel_err(dg.download(), dx_ref) < 1e-5 and rel_err(dx.download(), dg.download()) < 2e-6
    base = rng.normal(0, 1, (n, h, w, c)).astype(np.float32)
    dx.upload(base)
    ctx.call("ssdseg_bilinear_bwd", ctx.array(g), c, dx, c, n, h, w, c, fy, fx, 1)
    assert rel_err(dx.download(), dx_ref + base) < 1e-5
    # the gradient arriving as a slice of a wider (concat) gradient buffer, the result leaving into one
    gw = rng.normal(0, 1, (n * h * fy * w * fx, ldo)).astype(np.float32)
    gw[:, 4:4 + c] = g.reshape(-1, c)
    gbuf = ctx.array(gw)
    dxw = ctx.zeros((n * h * w, ldo))
    ctx.call("ssdseg_bilinear_bwd", gbuf.view(4, (gbuf.size - 4,)), ldo, dxw.view(4, (dxw.size - 4,)), ldo, n, h, w, c, fy, fx, 0)
    got = dxw.download()
    assert rel_err(got[:, 4:4 + c].reshape(dx_ref.shape), dx_ref) < 1e-5
    assert np.all(got[:, :4] == 0) and np.all(got[:, 4 + c:] == 0)


@pytest.mark.parametrize("n,h,w,c,f,ld", [(2, 6, 8, 8, 4, 8), (3, 5, 7, 12, 4, 20), (2, 9, 4, 8, 2, 12)])
def test_bilinear_fwd_padded(ctx, rng, n, h, w, c, f, ld):
    """the up-sampling written into the interior of a bordered tensor == the plain up-sampling, bit for bit; border and the
    channels beyond c untouched (x4 tile kernel and the general kernel)"""
    from ssdseglib import _hip as H
    x = rng.normal(0, 1, (n, h, w, c)).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(0, 0.3, c).astype(np.float32)
    v = H.view(ctx.array(x), ctx.array(sc), ctx.array(sh), O.ACT_RELU6)
    plain = ctx.empty((n, h * f, w * f, c))
    ctx.call("ssdseg_bilinear_fwd", v, c, plain, c, n, h, w, c, f, f)
    init = rng.normal(0, 1, (n, h * f + 2, w * f + 2, ld)).astype(np.float32)
    padded = ctx.array(init)
    ctx.call("ssdseg_bilinear_fwd_padded", v, c, padded, ld, n, h, w, c, f, f)
    got = padded.download()
    assert np.array_equal(got[:, 1:-1, 1:-1, :c], plain.download())
    want = init.copy()
    want[:, 1:-1, 1:-1, :c] = got[:, 1:-1, 1:-1, :c]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,h,w", [(2, 12, 16), (1, 17, 35), (3, 5, 3)])     # whole 16x16 tiles; ragged tiles; smaller than one tile
@pytest.mark.parametrize("f", [4, 8])
def test_mask_head(ctx, rng, monkeypatch, n, h, w, f):
    c = 4
    logits = rng.normal(0, 2, (n, h, w, c)).astype(np.float32)
    cls = rng.integers(0, c, (n, h * f, w * f))
    y = np.eye(c, dtype=np.float32)[cls]
    cw = np.array([0.05, 0.575, 0.135, 0.24], np.float32)
    up = O.bilinear_fwd(logits.astype(np.float64), f, f)
    p_ref = O.softmax(up)
    loss_ref, dp = O.cross_entropy_loss(y.astype(np.float64), p_ref, cw.astype(np.float64))
    dlogits_ref = O.bilinear_bwd(O.softmax_bwd(p_ref, dp * 0.5), f, f)
    cwh = (C.c_float * 4)(*cw)
    dl, dy = ctx.array(logits), ctx.array(y)
    prob, loss = ctx.empty(y.shape), ctx.empty(n)
    ctx.call("ssdseg_mask_head_fwd", dl, n, h, w, c, f, f, dy, cwh, prob, loss)
    assert np.abs(prob.download() - p_ref).max() < 2e-6
    assert rel_err(loss.download(), loss_ref) < 1e-5
    ctx.call("ssdseg_mask_head_fwd", dl, n, h, w, c, f, f, None, None, prob, None)     # inference: probabilities only
    assert np.abs(prob.download() - p_ref).max() < 2e-6
    g = ctx.empty(logits.shape)
    ctx.call("ssdseg_mask_head_bwd", dl, n, h, w, c, f, f, dy, cwh, 0.5, g)
    assert rel_err(g.download(), dlogits_ref) < 2e-5
    # the tile kernel (default for x4) computes every full-resolution dz once per block and walks each window in the order of the
    # one-thread-per-pixel kernel: bit-identical
    tile = g.download()
    monkeypatch.setenv("SSDSEG_MASK_BWD", "gather")
    ctx.call("ssdseg_mask_head_bwd", dl, n, h, w, c, f, f, dy, cwh, 0.5, g)
    if f == 4:
        np.testing.assert_array_equal(g.download(), tile)
    else:       # x8: the window of a pixel is folded by four threads (fixed order, but not the one-thread kernel's)
        assert rel_err(g.download(), dlogits_ref) < 2e-5 and rel_err(g.download(), tile) < 2e-6


@pytest.mark.parametrize("squared", [0, 1])
@pytest.mark.parametrize("n,h,w", [(2, 6, 8), (3, 30, 40)])
def test_mask_head_dice(ctx, rng, monkeypatch, n, h, w, squared):
    """the mask head trained with the reference's dice / dice_square losses (losses.py:175-264): loss and d loss / d logits vs the
    oracle (bilinear x4 -> softmax -> dice), tile and gather backward kernels bit-identical"""
    c, f = 4, 4
    logits = rng.normal(0, 2, (n, h, w, c)).astype(np.float32)
    cls = rng.integers(0, c, (n, h * f, w * f))
    y = np.eye(c, dtype=np.float32)[cls]
    cw = np.array([0.05, 0.575, 0.135, 0.24], np.float32)
    p_ref = O.softmax(O.bilinear_fwd(logits.astype(np.float64), f, f))
    loss_ref, dp = O.dice_loss_grad(y.astype(np.float64), p_ref, cw.astype(np.float64), squared=bool(squared))
    assert np.allclose(loss_ref, O.dice_loss(y.astype(np.float64), p_ref, cw.astype(np.float64), squared=bool(squared)), rtol=1e-12)
    dlogits_ref = O.bilinear_bwd(O.softmax_bwd(p_ref, dp * 0.5), f, f)
    cwh = (C.c_float * 4)(*cw)
    dl, dy = ctx.array(logits), ctx.array(y)
    prob, loss, coef = ctx.empty(y.shape), ctx.empty(n), ctx.empty((n, 8))
    ctx.call("ssdseg_mask_head_fwd_dice", dl, n, h, w, c, f, f, dy, cwh, squared, prob, loss, coef)
    assert np.abs(prob.download() - p_ref).max() < 2e-6
    assert rel_err(loss.download(), loss_ref) < 1e-5
    g = ctx.empty(logits.shape)
    ctx.call("ssdseg_mask_head_bwd_dice", dl, n, h, w, c, f, f, dy, coef, squared, 0.5, g)
    assert rel_err(g.download(), dlogits_ref) < 2e-5
    tile = g.download()
    monkeypatch.setenv("SSDSEG_MASK_BWD", "gather")
    ctx.call("ssdseg_mask_head_bwd_dice", dl, n, h, w, c, f, f, dy, coef, squared, 0.5, g)
    np.testing.assert_array_equal(g.download(), tile)
    # the standalone loss function (ssdseglib.losses.dice(w)(y_true, y_pred)) agrees with the fused head
    std = ctx.empty(n)
    ctx.call("ssdseg_dice_loss", dy, prob, n, h * f * w * f, c, cwh, squared, std)
    assert rel_err(std.download(), loss_ref) < 1e-5


def test_head_gather_and_softmax(ctx, rng):
    from ssdseglib import _hip as H
    b, hw, c, off, total = 3, 20, 24, 7, 200          # 20 cells x 6 boxes x 4 values per image, placed at anchor 7
    x, sc, sh, a = view_inputs(rng, (b, hw, c), O.ACT_RELU6)
    out = ctx.zeros((b, total, 4))
    ctx.call("ssdseg_head_gather", H.view(ctx.array(x), ctx.array(sc), ctx.array(sh), O.ACT_RELU6), out, b, hw * c, c, off * 4, total * 4, 0)
    got = out.download()
    ref = a.reshape(b, hw * c // 4, 4)
    assert np.abs(got[:, off:off + ref.shape[1]] - ref).max() < 1e-6
    assert np.all(got[:, :off] == 0) and np.all(got[:, off + ref.shape[1]:] == 0)
    back = ctx.empty((b, hw, c))
    ctx.call("ssdseg_head_gather", H.view(out), back, b, hw * c, c, off * 4, total * 4, 1)
    assert np.array_equal(back.download().reshape(ref.shape), got[:, off:off + ref.shape[1]])
    sm = ctx.empty((b, total, 4))
    ctx.call("ssdseg_softmax_rows", H.view(out), sm, b * total, 4)
    assert np.abs(sm.download() - O.softmax(got)).max() < 1e-6


@pytest.mark.parametrize("n,k", [(1000, 0), (1000, 1), (5000, 1234), (307200, 2700), (4096, 4095), (4096, 4096), (10, 50),
                                 (307200, 300001), (700001, 123457), (257, 5), (1, 1)])
def test_topk_mask_exact(ctx, rng, n, k):
    v = rng.exponential(1.0, n).astype(np.float32)
    v[rng.integers(0, n, n // 3)] = 0.0                     # plateau of ties at 0 (positives contribute 0 to the mining vector)
    v[rng.integers(0, n, n // 5)] = np.float32(0.6931472)   # plateau of ties at a positive value
    if n > 100:
        v[:7] = -1.5                                        # negative values order correctly too
    mask = ctx.empty(n, np.uint8)
    ctx.call("ssdseg_topk_mask", ctx.array(v), n, k, mask)
    assert np.array_equal(mask.download(), O.topk_mask(v, min(k, n)))


@pytest.mark.parametrize("n,k", [(307200, 1000), (5000, 4999), (300, 1)])
def test_topk_mask_all_ties(ctx, n, k):
    """every value equal: the selection is the k lowest indices, across block and thread boundaries of the tie pass"""
    v = np.full(n, 0.25, np.float32)
    mask = ctx.empty(n, np.uint8)
    ctx.call("ssdseg_topk_mask", ctx.array(v), n, k, mask)
    got = mask.download()
    assert got[:k].all() and not got[k:].any()


def make_det_case(rng, b, a, pos_frac=0.02):
    logits = rng.uniform(0, 6, (b, a, 4)).astype(np.float32)       # head outputs pass ReLU6 (quirk Q3)
    p = O.softmax(logits.astype(np.float64)).astype(np.float32)
    cls = np.where(rng.uniform(size=(b, a)) < pos_frac, rng.integers(1, 4, (b, a)), 0)
    y = np.eye(4, dtype=np.float32)[cls]
    yb = (rng.normal(0, 2, (b, a, 4)) * (cls > 0)[..., None]).astype(np.float32)
    pb = rng.uniform(0, 6, (b, a, 4)).astype(np.float32)
    return y, p, yb, pb


@pytest.mark.parametrize("b,a,pos_frac", [(4, 600, 0.03), (2, 9600, 0.01), (3, 500, 0.0), (2, 300, 0.6)])
def test_det_loss(ctx, rng, b, a, pos_frac):
    y, p, yb, pb = make_det_case(rng, b, a, pos_frac)
    conf_ref, dp_ref, keep_ref = O.confidence_loss(y, p)
    loc_ref, dloc_ref = O.localization_loss(yb, pb)
    scale = 1.0 / b
    dlogits_ref = O.softmax_bwd(p.astype(np.float64), dp_ref.astype(np.float64)) * scale
    conf, loc = ctx.empty(b), ctx.empty(b)
    dlog, dbox = ctx.empty((b, a, 4)), ctx.empty((b, a, 4))
    keep = ctx.empty(b * a, np.uint8)
    ctx.call("ssdseg_det_loss", ctx.array(y), ctx.array(p), ctx.array(yb), ctx.array(pb), b, a, 4, scale, conf, loc, dlog, dbox, keep)
    # the mining key is the correctly rounded float32 log on both sides (float(log(double(p))): DESIGN.md section 4), so the
    # selected set is held bit-exact
    got_keep = keep.download()
    assert np.array_equal(got_keep, keep_ref), f"{(got_keep != keep_ref).sum()} hard-negative decisions differ from the oracle"
    assert rel_err(dlog.download(), dlogits_ref) < 2e-5 or np.abs(dlogits_ref).max() == 0
    assert rel_err(conf.download(), conf_ref) < 1e-5 or np.abs(conf_ref).max() == 0
    assert rel_err(loc.download(), loc_ref) < 1e-5 or np.abs(loc_ref).max() == 0
    assert np.abs(dbox.download() - dloc_ref * scale).max() < 1e-6


def synthetic_gt(rng, b, gmax, hw=(480, 640)):
    gt = np.zeros((b, gmax, 5), np.float32)
    cnt = np.zeros(b, np.int32)
    for i in range(b):
        g = int(rng.integers(0, gmax + 1)) if i else gmax
        cnt[i] = g
        w = np.exp(rng.uniform(np.log(24), np.log(400), g)); h = np.exp(rng.uniform(np.log(24), np.log(400), g))
        x0 = rng.uniform(0, np.maximum(hw[1] - w, 1)); y0 = rng.uniform(0, np.maximum(hw[0] - h, 1))
        gt[i, :g] = np.stack([rng.integers(1, 4, g), x0, y0, np.minimum(x0 + w, hw[1] - 1), np.minimum(y0 + h, hw[0] - 1)], axis=1)
    return gt, cnt


def test_encode_targets_exact(ctx, rng, golden_dir):
    d = np.load(f"{golden_dir}/anchors_nb03.npz")
    anchors = d["corners"]
    b, gmax = 6, 8
    gt, cnt = synthetic_gt(rng, b, gmax)
    cnt[1] = 0                                              # image without objects
    gt[2, 1] = gt[2, 0]                                     # duplicated ground truth: arg-max ties
    gt[3, 0, 1:] = anchors[4321]                            # a box that coincides with an anchor (IoU == 1)
    stds = (0.1, 0.1, 0.2, 0.2)
    labels, boxes, match = ctx.empty((b, 9600, 4)), ctx.empty((b, 9600, 4)), ctx.empty((b, 9600), np.int32)
    ctx.call("ssdseg_encode_targets", ctx.array(anchors), 9600, ctx.array(gt), ctx.array(cnt), b, gmax, 4, 0.525, (C.c_float * 4)(*stds),
             labels, boxes, match)
    L, B, M = labels.download(), boxes.download(), match.download()
    for i in range(b):
        l_ref, b_ref, m_ref = O.encode_targets(anchors, gt[i, :cnt[i]], 4, 0.525, stds)
        assert np.array_equal(M[i], m_ref), f"image {i}: matched ground-truth indices differ"
        assert np.array_equal(L[i], l_ref)
        assert np.abs(B[i] - b_ref).max() < 1e-5
    assert (M[1] == -1).all() and (L[1, :, 0] == 1).all()
    assert M[3, 4321] == 0
    # encode -> decode round trip (datacoder.py:266-269 <-> :371-374)
    cent = d["centroids"]
    dec = O.decode_to_centroids_gt(B[0], cent, stds)
    pos = np.nonzero(M[0] >= 0)[0]
    g = gt[0, M[0][pos]]
    want = np.stack([(g[:, 3] + g[:, 1]) / 2, (g[:, 4] + g[:, 2]) / 2, g[:, 3] - g[:, 1] + 1, g[:, 4] - g[:, 2] + 1], axis=1)
    assert np.abs(dec[pos] - want).max() < 2e-2


@pytest.mark.parametrize("iou_thr,score_thr", [(0.025, 0.725), (0.5, 0.05), (0.3, 0.9999)])
def test_decode_and_combined_nms_exact(ctx, rng, golden_dir, iou_thr, score_thr):
    d = np.load(f"{golden_dir}/anchors_nb03.npz")
    cent = d["centroids"]
    b, a, c = 4, 9600, 4
    offsets = rng.uniform(0, 6, (b, a, 4)).astype(np.float32)
    probs = O.softmax((3 * rng.normal(0, 1, (b, a, c))).astype(np.float32))
    probs[3, :, 1:] = 0.0                                        # image with only background-class candidates
    stds = (0.1, 0.1, 0.2, 0.2)
    corners = ctx.empty((b, a, 4))
    ctx.call("ssdseg_decode_boxes", ctx.array(offsets), ctx.array(cent), b, a, (C.c_float * 4)(*stds), corners)
    cr = corners.download()
    assert rel_err(cr, O.decode_to_corners_pred(offsets, cent, stds)) < 1e-5
    out, valid = ctx.empty((b, 10, 6)), ctx.empty(b, np.int32)
    ctx.call("ssdseg_combined_nms", corners, ctx.array(probs), b, a, c, 4, 10, iou_thr, score_thr, out, valid)
    ref_out, ref_valid = O.combined_nms(cr, probs, 4, 10, iou_thr, score_thr)      # same decoded boxes on both sides
    assert np.array_equal(valid.download(), ref_valid)
    assert np.array_equal(out.download(), ref_out)


def test_seg_suppress(ctx, rng):
    n, hw, c, rows = 2, 4800, 4, 500
    mask = O.softmax(rng.normal(0, 1, (n, hw, c)).astype(np.float32))
    mask[..., 2] = 0.0                                           # class 2 never wins anywhere in the batch
    mask[0, :10] = 0.25                                          # exact ties -> first index (class 0)
    probs = rng.uniform(0, 1, (rows, c)).astype(np.float32)
    out = ctx.empty((rows, c))
    ctx.call("ssdseg_seg_suppress", ctx.array(mask), n * hw, c, ctx.array(probs), rows, out)
    assert np.array_equal(out.download(), O.seg_suppress(mask, probs))
    assert np.all(out.download()[:, 2] == 0)


def test_maxpool_shuffle_actbwd_dice(ctx, rng, monkeypatch):
    from ssdseglib import _hip as H
    n, h, w, c = 2, 15, 20, 24
    x = rng.normal(0, 1, (n, h, w, c)).astype(np.float32)
    x[0, 3:6, 3:6, :] = 1.5                                      # ties inside windows
    ref = O.maxpool3x3s2_fwd(x)
    out = ctx.empty(ref.shape)
    dx_ = ctx.array(x)
    ctx.call("ssdseg_maxpool3x3s2_fwd", H.view(dx_), out, n, h, w, c)
    assert np.array_equal(out.download(), ref)
    g = rng.normal(0, 1, ref.shape).astype(np.float32)
    dx = ctx.empty(x.shape)
    ctx.call("ssdseg_maxpool3x3s2_bwd", H.view(dx_), ctx.array(g), dx, n, h, w, c)
    assert np.abs(dx.download() - O.maxpool3x3s2_bwd(x, g)).max() < 1e-6
    # the two-pass form (winner codes, default) and the one-pass window scan walk the windows in the same order: bit-identical;
    # even image sizes pad differently (SAME: (0, 1)) and a view on the input
    for (hh, ww, view) in ((15, 20, False), (16, 22, True), (1, 1, False)):
        xx = rng.normal(0, 1, (n, hh, ww, c)).astype(np.float32)
        sc, shf = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(0, 1, c).astype(np.float32)
        a = np.maximum(xx * sc + shf, 0).astype(np.float32) if view else xx
        v = H.view(ctx.array(xx), ctx.array(sc), ctx.array(shf), O.ACT_RELU) if view else H.view(ctx.array(xx))
        gg = rng.normal(0, 1, O.maxpool3x3s2_fwd(a).shape).astype(np.float32)
        d1, d2 = ctx.empty(xx.shape), ctx.empty(xx.shape)
        monkeypatch.delenv("SSDSEG_MAXPOOL_BWD", raising=False)
        ctx.call("ssdseg_maxpool3x3s2_bwd", v, ctx.array(gg), d1, n, hh, ww, c)
        monkeypatch.setenv("SSDSEG_MAXPOOL_BWD", "scan")
        ctx.call("ssdseg_maxpool3x3s2_bwd", v, ctx.array(gg), d2, n, hh, ww, c)
        monkeypatch.delenv("SSDSEG_MAXPOOL_BWD", raising=False)
        assert np.array_equal(d1.download(), d2.download())
        assert np.abs(d1.download() - O.maxpool3x3s2_bwd(a, gg)).max() < 1e-5
    sh = ctx.empty(x.shape)
    ctx.call("ssdseg_channel_shuffle", H.view(dx_), c, sh, c, n * h * w, c, 2, 0)
    assert np.array_equal(sh.download(), O.channel_shuffle(x, 2))
    back = ctx.empty(x.shape)
    ctx.call("ssdseg_channel_shuffle", H.view(sh), c, back, c, n * h * w, c, 2, 1)
    assert np.array_equal(back.download(), x)
    sc, sf = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(0, 0.3, c).astype(np.float32)
    ctx.call("ssdseg_channel_shuffle", H.view(dx_, ctx.array(sc), ctx.array(sf), O.ACT_RELU), c, sh, c, n * h * w, c, 2, 0)   # fused BN + ReLU
    assert np.abs(sh.download() - O.channel_shuffle(np.maximum(x * sc + sf, 0), 2)).max() < 1e-6
    gg = ctx.array(g_full := rng.normal(0, 1, x.shape).astype(np.float32))
    ctx.call("ssdseg_act_bwd", gg, c, dx_, c, n * h * w, c, O.ACT_RELU)
    assert np.array_equal(gg.download(), g_full * (x > 0))
    y = np.eye(4, dtype=np.float32)[rng.integers(0, 4, (n, 300))]
    p = O.softmax(rng.normal(0, 1, (n, 300, 4)).astype(np.float32))
    cw = (0.05, 0.575, 0.135, 0.24)
    for sq in (0, 1):
        loss = ctx.empty(n)
        ctx.call("ssdseg_dice_loss", ctx.array(y), ctx.array(p), n, 300, 4, (C.c_float * 4)(*cw), sq, loss)
        assert rel_err(loss.download(), O.dice_loss(y[:, :, None], p[:, :, None], cw, bool(sq))) < 1e-5


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 6, 64, 80, 72), (1, 16, 32, 304, 256)])
def test_conv3x3_saved_input_pair_equals_plain_entry_points(ctx, rng, monkeypatch, n, h, w, cin, cout):
    """ssdseg_conv3x3_fwd_saved / _bwd_weight_saved (the forward leaves act(BN(x)) in a zero-bordered copy, the Winograd weight
    gradient reads it again): same outputs, BatchNorm partial sums and weight gradient as the plain entry points, and the oracle's"""
    import ctypes as C
    from ssdseglib import _hip as H
    monkeypatch.setenv("SSDSEG_CONV3_WINOGRAD", "1")
    monkeypatch.setenv("SSDSEG_CONV3_NARROW", "0")
    act = O.ACT_RELU6
    x, sc, sh, a = view_inputs(rng, (n, h, w, cin), act)
    wgt = (rng.normal(0, 1, (3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    dy = rng.normal(0, 1, (n, h, w, cout)).astype(np.float32)
    need = C.c_longlong()
    assert ctx.lib.ssdseg_conv3x3_saved_floats(n, h, w, cin, cout, C.byref(need)) == 0
    assert need.value == n * (h + 2) * (w + 2) * cin
    dx_, dsc, dsh, dw_, ddy = ctx.array(x), ctx.array(sc), ctx.array(sh), ctx.array(wgt), ctx.array(dy)
    nparts = ctx.parts("ssdseg_conv3x3_parts", n, h, w, cin, cout)
    y0, y1 = ctx.empty((n, h, w, cout)), ctx.empty((n, h, w, cout))
    st0, st1 = ctx.empty((nparts, 2, cout)), ctx.empty((nparts, 2, cout))
    g0, g1 = ctx.empty(wgt.shape), ctx.empty(wgt.shape)
    xs = ctx.empty(need.value)
    xs.upload(np.full(need.value, np.nan, np.float32))                       # every element, borders included, must be written
    ctx.call("ssdseg_conv3x3_fwd", H.view(dx_, dsc, dsh, act), cin, dw_, y0, n, h, w, cin, cout, st0)
    ctx.call("ssdseg_conv3x3_fwd_saved", H.view(dx_, dsc, dsh, act), cin, dw_, y1, n, h, w, cin, cout, st1, xs)
    saved = xs.download().reshape(n, h + 2, w + 2, cin)
    assert np.abs(saved[:, 1:-1, 1:-1] - a).max() < 2e-6 * np.abs(a).max()   # act(scale * x + shift) (one fma on the device, mul + add in NumPy)
    assert not saved[:, 0].any() and not saved[:, -1].any() and not saved[:, :, 0].any() and not saved[:, :, -1].any()
    y_ref = O.conv2d_fwd(a.astype(np.float64), wgt.astype(np.float64))
    assert rel_err(y1.download(), y_ref) < 2e-5
    assert rel_err(y1.download(), y0.download().astype(np.float64)) < 5e-6   # view applied before vs while staging: same values, same sums
    assert rel_err(st1.download().sum(0), st0.download().sum(0).astype(np.float64)) < 1e-5
    ctx.call("ssdseg_conv3x3_bwd_weight", H.view(dx_, dsc, dsh, act), cin, H.gview(ddy), g0, n, h, w, cin, cout)
    ctx.call("ssdseg_conv3x3_bwd_weight_saved", xs, ddy, g1, n, h, w, cin, cout)
    np.testing.assert_array_equal(g1.download(), g0.download())              # the same kernel on the same padded tensor
    _, dw_ref, _ = O.conv2d_bwd(a.astype(np.float64), wgt.astype(np.float64), dy.astype(np.float64))
    assert rel_err(g1.download(), dw_ref) < 5e-5
    # a shape the Winograd pair does not take reports no saved tensor and refuses the call
    assert ctx.lib.ssdseg_conv3x3_saved_floats(1, 5, 5, 24, 8, C.byref(need)) == 0 and need.value == 0
    with pytest.raises(H.SsdsegError):
        ctx.call("ssdseg_conv3x3_fwd_saved", H.view(dx_), 24, dw_, y1, 1, 5, 5, 24, 8, None, xs)
