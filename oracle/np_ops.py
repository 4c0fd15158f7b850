"""CPU oracle (NumPy) for the ssdseglib hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package; the product
(`ssdseglib/` + libssdseg_hip.so) never does and fails loudly without its HIP library.

What this is: a restatement of the arithmetic the reference executes on its hot path.  The reference is Python on
TensorFlow 2.13 / Keras (requirements.txt:1); TensorFlow is NOT installable here (no network), so the conv / BN /
resize / NMS arithmetic below restates the *published* TF/Keras semantics at the reference's
call sites, each function citing the reference file:line it follows.

Pinning status (see DESIGN.md "Oracle"):
  * anchors (boxes.py)      -- pinned bit-exact by executing the reference's NumPy-only boxes.py (tests/golden/anchors_*.npz)
  * graph structure         -- pinned by the reference's own `model.summary()` output (tests/golden/nb03_model_summary.json)
  * conv/BN/resize/softmax  -- cross-checked against torch-CPU (independent implementation + autograd) in tests;
                               TF itself: PARITY UNPINNED (the reference ships no tests / vectors for these)
  * encode / losses / NMS   -- PARITY UNPINNED against TF; pinned only by algebraic identities (encode<->decode
                               round trip, loss==0 cases) and hand-worked vectors in tests/golden/

All functions take/return NHWC arrays; `dt` selects float32 (default, what TF computes in) or float64 (for
finite-difference gradient checks).
"""
from __future__ import annotations

import numpy as np

ACT_NONE, ACT_RELU, ACT_RELU6, ACT_ZERO = 0, 1, 2, 3


# --------------------------------------------------------------------------------------------- geometry
def same_pad(size, k, s, d=1):
    """TF SAME padding (App. B.1): out=ceil(in/s); before=total//2, after=total-before."""
    out = -(-size // s)
    keff = (k - 1) * d + 1
    total = max((out - 1) * s + keff - size, 0)
    return out, total // 2, total - total // 2


def _pad_nhwc(x, k, s, d):
    n, h, w, c = x.shape
    ho, pt, pb = same_pad(h, k, s, d)
    wo, pl, pr = same_pad(w, k, s, d)
    xp = np.zeros((n, h + pt + pb, w + pl + pr, c), dtype=x.dtype)
    xp[:, pt:pt + h, pl:pl + w, :] = x
    return xp, ho, wo, pt, pl


def _tap(xp, kh, kw, ho, wo, s, d):
    return xp[:, kh * d: kh * d + (ho - 1) * s + 1: s, kw * d: kw * d + (wo - 1) * s + 1: s, :]


# --------------------------------------------------------------------------------------------- activations
def act_fwd(z, act):
    """Keras ReLU(max_value) (App. B.4; models.py:67,90; blocks.py:30 with relu_max_value)."""
    if act == ACT_RELU:
        return np.maximum(z, 0)
    if act == ACT_RELU6:
        return np.minimum(np.maximum(z, 0), 6)
    if act == ACT_ZERO:
        return np.zeros_like(z)
    return z


def act_mask(z, act):
    if act == ACT_RELU:
        return (z > 0).astype(z.dtype)
    if act == ACT_RELU6:
        return ((z > 0) & (z < 6)).astype(z.dtype)
    if act == ACT_ZERO:
        return np.zeros_like(z)
    return np.ones_like(z)


def rescale(x, scale=1.0 / 127.5, offset=-1.0):
    """Rescaling layer (models.py:187,622)."""
    return x * np.asarray(scale, x.dtype) + np.asarray(offset, x.dtype)


# --------------------------------------------------------------------------------------------- convolutions
def conv2d_fwd(x, w, stride=1, dilation=1, bias=None):
    """Conv2D SAME, NHWC x HWIO (models.py:65,110,628; blocks.py:28,58,70,109,117,127)."""
    k = w.shape[0]
    xp, ho, wo, _, _ = _pad_nhwc(x, k, stride, dilation)
    n = x.shape[0]
    y = np.zeros((n, ho, wo, w.shape[3]), dtype=x.dtype)
    for kh in range(k):
        for kw in range(k):
            y += _tap(xp, kh, kw, ho, wo, stride, dilation) @ w[kh, kw]
    if bias is not None:
        y += bias
    return y


def conv2d_bwd(x, w, dy, stride=1, dilation=1):
    """-> (dx, dw, dbias)."""
    k = w.shape[0]
    xp, ho, wo, pt, pl = _pad_nhwc(x, k, stride, dilation)
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    n, h, wd, _ = x.shape
    for kh in range(k):
        for kw in range(k):
            tap = _tap(xp, kh, kw, ho, wo, stride, dilation)
            dw[kh, kw] = np.tensordot(tap, dy, axes=([0, 1, 2], [0, 1, 2]))
            _tap(dxp, kh, kw, ho, wo, stride, dilation)[...] += dy @ w[kh, kw].T
    dx = dxp[:, pt:pt + h, pl:pl + wd, :]
    return dx, dw, dy.sum(axis=(0, 1, 2))


def dwconv_fwd(x, w, stride=1, dilation=1):
    """DepthwiseConv2D 3x3 SAME, depth multiplier 1; w: (3,3,C) (models.py:88; depthwise half of SeparableConv2D)."""
    xp, ho, wo, _, _ = _pad_nhwc(x, 3, stride, dilation)
    y = np.zeros((x.shape[0], ho, wo, x.shape[3]), dtype=x.dtype)
    for kh in range(3):
        for kw in range(3):
            y += _tap(xp, kh, kw, ho, wo, stride, dilation) * w[kh, kw]
    return y


def dwconv_bwd(x, w, dy, stride=1, dilation=1):
    xp, ho, wo, pt, pl = _pad_nhwc(x, 3, stride, dilation)
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    n, h, wd, _ = x.shape
    for kh in range(3):
        for kw in range(3):
            dw[kh, kw] = (_tap(xp, kh, kw, ho, wo, stride, dilation) * dy).sum(axis=(0, 1, 2))
            _tap(dxp, kh, kw, ho, wo, stride, dilation)[...] += dy * w[kh, kw]
    return dxp[:, pt:pt + h, pl:pl + wd, :], dw


# --------------------------------------------------------------------------------------------- batch norm
def bn_train_fwd(y, gamma, beta, eps=1e-3):
    """BatchNormalization(training=True): batch mean / biased variance over (N,H,W) (App. B.3).
    -> z, cache.  cache['scale'], cache['shift'] are the fused per-channel affine."""
    axes = tuple(range(y.ndim - 1))
    mean = y.mean(axis=axes, dtype=np.float64)
    var = y.astype(np.float64).var(axis=axes)
    invstd = 1.0 / np.sqrt(var + eps)
    dt = y.dtype
    scale = (gamma.astype(np.float64) * invstd).astype(dt)
    shift = (beta.astype(np.float64) - mean * gamma.astype(np.float64) * invstd).astype(dt)
    z = y * scale + shift
    count = y.size // y.shape[-1]
    return z, dict(mean=mean.astype(dt), var=var.astype(dt), invstd=invstd.astype(dt), scale=scale, shift=shift, count=count)


def bn_moving_update(moving_mean, moving_var, cache, momentum=0.99):
    """Keras moving statistics; the fused kernel feeds the Bessel-corrected variance (App. B.3)."""
    n = cache["count"]
    unbiased = cache["var"] * (n / (n - 1.0)) if n > 1 else cache["var"]
    mm = moving_mean * momentum + cache["mean"] * (1.0 - momentum)
    mv = moving_var * momentum + unbiased * (1.0 - momentum)
    return mm.astype(moving_mean.dtype), mv.astype(moving_var.dtype)


def bn_infer_affine(gamma, beta, moving_mean, moving_var, eps=1e-3):
    invstd = 1.0 / np.sqrt(moving_var.astype(np.float64) + eps)
    scale = gamma * invstd
    return scale.astype(gamma.dtype), (beta - moving_mean * scale).astype(gamma.dtype)


def bn_train_bwd(dz, y, gamma, cache):
    """-> (dy, dgamma, dbeta) for z = gamma*(y-mean)*invstd + beta with batch statistics."""
    axes = tuple(range(y.ndim - 1))
    m = cache["count"]
    xhat = (y - cache["mean"]) * cache["invstd"]
    dbeta = dz.sum(axis=axes, dtype=np.float64)
    dgamma = (dz * xhat).sum(axis=axes, dtype=np.float64)
    dt = y.dtype
    dy = cache["scale"] * (dz - (dbeta / m).astype(dt) - xhat * (dgamma / m).astype(dt))
    return dy.astype(dt), dgamma.astype(dt), dbeta.astype(dt)


# --------------------------------------------------------------------------------------------- pooling / resize
def gap_fwd(x):
    """GlobalAveragePooling2D(keepdims=True) (blocks.py:57)."""
    return x.mean(axis=(1, 2), keepdims=True, dtype=np.float64).astype(x.dtype)


def gap_bwd(g, h, w):
    return np.broadcast_to(g / np.asarray(h * w, g.dtype), (g.shape[0], h, w, g.shape[3])).copy()


def _bilinear_axis(in_size, factor, dt):
    """half-pixel centres, no align-corners: src=(dst+0.5)/factor-0.5 clamped to [0,in-1] (App. B.5)."""
    out = in_size * factor
    src = (np.arange(out, dtype=np.float64) + 0.5) / factor - 0.5
    src = np.clip(src, 0.0, in_size - 1)
    i0 = np.floor(src).astype(np.int64)
    i1 = np.minimum(i0 + 1, in_size - 1)
    f = (src - i0).astype(dt)
    return i0, i1, f


def bilinear_fwd(x, fy, fx):
    """UpSampling2D(size=(fy,fx), interpolation='bilinear') == tf.image.resize half-pixel (blocks.py:61,104,129)."""
    n, h, w, c = x.shape
    y0, y1, wy = _bilinear_axis(h, fy, x.dtype)
    x0, x1, wx = _bilinear_axis(w, fx, x.dtype)
    top = x[:, y0][:, :, x0] * (1 - wx)[None, None, :, None] + x[:, y0][:, :, x1] * wx[None, None, :, None]
    bot = x[:, y1][:, :, x0] * (1 - wx)[None, None, :, None] + x[:, y1][:, :, x1] * wx[None, None, :, None]
    return top * (1 - wy)[None, :, None, None] + bot * wy[None, :, None, None]


def bilinear_bwd(g, fy, fx):
    n, ho, wo, c = g.shape
    h, w = ho // fy, wo // fx
    y0, y1, wy = _bilinear_axis(h, fy, g.dtype)
    x0, x1, wx = _bilinear_axis(w, fx, g.dtype)
    # rows
    tmp = np.zeros((n, h, wo, c), dtype=g.dtype)
    np.add.at(tmp, (slice(None), y0), g * (1 - wy)[None, :, None, None])
    np.add.at(tmp, (slice(None), y1), g * wy[None, :, None, None])
    dx = np.zeros((n, h, w, c), dtype=g.dtype)
    np.add.at(dx, (slice(None), slice(None), x0), tmp * (1 - wx)[None, None, :, None])
    np.add.at(dx, (slice(None), slice(None), x1), tmp * wx[None, None, :, None])
    return dx


def maxpool3x3s2_fwd(x):
    """MaxPooling2D(3, strides=2, padding='same'): padded cells never win (models.py:629)."""
    n, h, w, c = x.shape
    ho, pt, pb = same_pad(h, 3, 2)
    wo, pl, pr = same_pad(w, 3, 2)
    xp = np.full((n, h + pt + pb, w + pl + pr, c), -np.inf, dtype=x.dtype)
    xp[:, pt:pt + h, pl:pl + w] = x
    out = np.full((n, ho, wo, c), -np.inf, dtype=x.dtype)
    for kh in range(3):
        for kw in range(3):
            out = np.maximum(out, _tap(xp, kh, kw, ho, wo, 2, 1))
    return out


def maxpool3x3s2_bwd(x, g):
    """gradient of MaxPooling2D(3, 2, 'same'): routed to the first maximum of each window (row-major scan)."""
    n, h, w, c = x.shape
    ho, pt, pb = same_pad(h, 3, 2)
    wo, pl, pr = same_pad(w, 3, 2)
    xp = np.full((n, h + pt + pb, w + pl + pr, c), -np.inf, dtype=x.dtype)
    xp[:, pt:pt + h, pl:pl + w] = x
    dxp = np.zeros_like(xp)
    taps = np.stack([_tap(xp, kh, kw, ho, wo, 2, 1) for kh in range(3) for kw in range(3)], axis=0)  # (9, n, ho, wo, c)
    first = taps.argmax(axis=0)                                                                       # first maximum
    for t in range(9):
        kh, kw = divmod(t, 3)
        _tap(dxp, kh, kw, ho, wo, 2, 1)[...] += g * (first == t)
    return dxp[:, pt:pt + h, pl:pl + w]


def channel_shuffle(x, groups=2):
    """Reshape(h,w,g,c/g) -> Permute(1,2,4,3) -> Reshape (models.py:497-503)."""
    n, h, w, c = x.shape
    return x.reshape(n, h, w, groups, c // groups).transpose(0, 1, 2, 4, 3).reshape(n, h, w, c)


# --------------------------------------------------------------------------------------------- softmax + losses
EPS = 1e-7  # tf.keras.backend.epsilon()


def softmax(x):
    """Softmax(axis=-1) (blocks.py:130; models.py:259)."""
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def softmax_bwd(p, dp):
    return p * (dp - (dp * p).sum(axis=-1, keepdims=True))


def _clipped_log(p):
    dt = p.dtype
    lo, hi = np.asarray(EPS, dt), np.asarray(1.0, dt) - np.asarray(EPS, dt)
    inside = ((p >= lo) & (p <= hi)).astype(dt)   # clip_by_value gradient (App. B.6)
    # float32 input: the logarithm is taken in double and rounded once -- the correctly rounded float32 log.  TF's own
    # float32 log (Eigen) is a <= 1 ulp approximation; which last bit it returns is not knowable here, and the hard-negative
    # selection ranks near-equal background losses by exactly that bit.  Defining the key as the correctly rounded value makes
    # the selection reproducible across math libraries (the device kernel computes float(log(double(p))) too).
    clipped = np.clip(p, lo, hi)
    return np.log(clipped.astype(np.float64)).astype(dt), inside


def cross_entropy_loss(y_true, p, class_weights):
    """losses.cross_entropy(w) (losses.py:294-305) -> loss (B,), dL_b/dp."""
    logp, inside = _clipped_log(p)
    w = np.asarray(class_weights, p.dtype)
    per_class = -(y_true * logp).sum(axis=(1, 2))
    loss = (per_class * w).sum(axis=-1)
    dp = -(y_true * w) / np.clip(p, EPS, 1 - EPS) * inside
    return loss, dp


def dice_loss(y_true, p, class_weights, squared=False):
    """losses.dice / dice_square (losses.py:204-216, 250-262)."""
    w = np.asarray(class_weights, p.dtype)
    inter = (y_true * p).sum(axis=(1, 2))
    total = (y_true ** 2 + p ** 2).sum(axis=(1, 2)) if squared else (y_true + p).sum(axis=(1, 2))
    eps = np.asarray(EPS, p.dtype)
    return ((1.0 - (2.0 * inter + eps) / (total + eps)) * w).sum(axis=-1)


def dice_loss_grad(y_true, p, class_weights, squared=False):
    """losses.dice / dice_square (losses.py:204-216, 250-262) -> loss (B,), dL_b/dp: with I = sum y p and T = sum (y + p)
    [sum (y^2 + p^2)] over the pixels of an image, d/dp [1 - (2 I + eps) / (T + eps)] = -2 y / (T + eps) + (2 I + eps) / (T + eps)^2 * dT/dp,
    dT/dp = 1 [2 p]."""
    w = np.asarray(class_weights, p.dtype)
    inter = (y_true * p).sum(axis=(1, 2), keepdims=True)
    total = (y_true ** 2 + p ** 2).sum(axis=(1, 2), keepdims=True) if squared else (y_true + p).sum(axis=(1, 2), keepdims=True)
    eps = np.asarray(EPS, p.dtype)
    loss = ((1.0 - (2.0 * inter + eps) / (total + eps)) * w).sum(axis=(1, 2, 3))
    dt = 2.0 * p if squared else np.ones_like(p)
    dp = w * (-2.0 * y_true / (total + eps) + (2.0 * inter + eps) / (total + eps) ** 2 * dt)
    return loss, dp


def localization_loss(y_true, y_pred):
    """losses.localization_loss (losses.py:21-49) -> loss (B,), dL_b/dy_pred."""
    dt = y_pred.dtype
    notbg = (np.abs(y_true).sum(axis=-1) > 0).astype(dt)
    err = y_true - y_pred
    a = np.abs(err)
    sl1 = np.where(a < 1.0, err * err * 0.5, a - 0.5).sum(axis=-1) * notbg
    npos = np.maximum(notbg.sum(axis=-1), 1.0)
    loss = sl1.sum(axis=-1) / npos
    d = np.where(a < 1.0, -err, -np.sign(err)) * notbg[..., None] / npos[:, None, None]
    return loss.astype(dt), d.astype(dt)


def topk_mask(values, k):
    """tf.math.top_k selection (losses.py:131; App. B.8): larger first, lower index first among equals."""
    n = values.shape[0]
    mask = np.zeros(n, dtype=np.uint8)
    if k > 0:
        order = np.lexsort((np.arange(n), -values.astype(np.float64)))  # stable: value desc, index asc
        mask[order[:k]] = 1
    return mask


def confidence_loss(y_true, p):
    """losses.confidence_loss (losses.py:70-172): batch-global 3:1 hard-negative mining.
    -> loss (B,), dL_b/dp (B,A,C), keep mask (B*A,) uint8."""
    dt = p.dtype
    b, a, c = p.shape
    is_bg = y_true[:, :, 0]
    not_bg = np.abs(is_bg - 1.0)
    n_bg = int(np.count_nonzero(is_bg))
    n_pos = int(np.count_nonzero(not_bg))
    logp, inside = _clipped_log(p)
    ce = -(y_true * logp).sum(axis=-1)
    pos_loss = (ce * not_bg).sum(axis=-1)
    npos_b = not_bg.sum(axis=-1)
    keep = np.zeros(b * a, dtype=np.uint8)
    if n_bg == 0:
        bg_loss = np.zeros_like(pos_loss)
    else:
        k = min(3 * n_pos, n_bg)
        bg = (ce * is_bg).reshape(-1)
        keep = topk_mask(bg, k)
        bg_loss = (bg * keep.astype(dt)).reshape(b, a).sum(axis=-1)
    denom = np.maximum(npos_b, 1.0)
    loss = (pos_loss + bg_loss) / denom
    sel = not_bg + is_bg * keep.reshape(b, a).astype(dt)
    dp = -(y_true / np.clip(p, EPS, 1 - EPS)) * inside * (sel / denom[:, None])[..., None]
    return loss.astype(dt), dp.astype(dt), keep


# --------------------------------------------------------------------------------------------- anchors: encode / decode / NMS
def encode_targets(anchors_corners, gt, num_classes, iou_threshold, stds):
    """DataEncoderDecoder._encode_ground_truth_labels_boxes (datacoder.py:205-300), one sample.
    anchors_corners (A,4) xmin,ymin,xmax,ymax; gt (G,5) label,xmin,ymin,xmax,ymax (all float32).
    -> labels (A,C) one-hot, boxes (A,4) offsets, match (A,) int32 gt index or -1."""
    f = np.float32
    ax0, ay0, ax1, ay1 = (anchors_corners[:, i].astype(f) for i in range(4))
    a_n = ax0.shape[0]
    labels = np.zeros((a_n, num_classes), f)
    labels[:, 0] = 1.0
    boxes = np.zeros((a_n, 4), f)
    match = np.full(a_n, -1, np.int32)
    g_n = gt.shape[0]
    if g_n == 0:
        return labels, boxes, match
    gl = gt[:, 0].astype(np.int64)
    gx0, gy0, gx1, gy1 = (gt[:, i].astype(f) for i in range(1, 5))
    one = f(1.0)
    area_a = ((ay1 - ay0 + one) * (ax1 - ax0 + one))[:, None]                                  # :112
    area_g = (gx1 - gx0 + one) * (gy1 - gy0 + one)                                              # :206
    ix0 = np.maximum(ax0[:, None], gx0[None, :]); iy0 = np.maximum(ay0[:, None], gy0[None, :])  # :210-213
    ix1 = np.minimum(ax1[:, None], gx1[None, :]); iy1 = np.minimum(ay1[:, None], gy1[None, :])
    inter = np.maximum(f(0), ix1 - ix0 + one) * np.maximum(f(0), iy1 - iy0 + one)               # :216
    iou = inter / (area_a + area_g[None, :] - inter)                                            # :220
    # step 1: best anchor of every gt with IoU > 0 (:230-231); step 2: best gt of every anchor above threshold (:236-241)
    rows = [(int(np.argmax(iou[:, g])), g) for g in range(g_n) if iou[:, g].max() > 0.0]
    best_g = np.argmax(iou, axis=1)
    best_v = iou.max(axis=1)
    rows += [(int(d), int(best_g[d])) for d in np.nonzero(best_v > f(iou_threshold))[0]]
    seen, uniq = set(), []
    for r in rows:                                                                              # UniqueV2 keeps first occurrences (:244)
        if r not in seen:
            seen.add(r)
            uniq.append(r)
    sx, sy, sw, sh = (f(s) for s in stds)
    for d, g in uniq:                                                                           # sequential scatter: last row wins (:286)
        acx = (ax1[d] + ax0[d]) / f(2); acy = (ay1[d] + ay0[d]) / f(2)
        aw = ax1[d] - ax0[d] + one; ah = ay1[d] - ay0[d] + one
        gcx = (gx1[g] + gx0[g]) / f(2); gcy = (gy1[g] + gy0[g]) / f(2)
        gw = gx1[g] - gx0[g] + one; gh = gy1[g] - gy0[g] + one
        labels[d] = 0.0
        labels[d, gl[g]] = 1.0
        boxes[d] = ((gcx - acx) / aw / sx, (gcy - acy) / ah / sy,
                    np.log(gw / aw + one) / sw, np.log(gh / ah + one) / sh)                     # :266-269
        match[d] = g
    return labels, boxes, match


def decode_to_corners_pred(offsets, anchors_centroids, stds):
    """layers.DecodeBoxesCentroidsOffsets.call (layers.py:58-79) -> (..., A, 4) as (ymin, xmin, ymax, xmax)."""
    dt = offsets.dtype
    acx, acy, aw, ah = (anchors_centroids[:, i].astype(dt) for i in range(4))
    sx, sy, sw, sh = (np.asarray(s, dt) for s in stds)
    cx = offsets[..., 0] * sx * aw + acx
    cy = offsets[..., 1] * sy * ah + acy
    w = (np.exp(offsets[..., 2] * sw) - 1) * aw
    h = (np.exp(offsets[..., 3] * sh) - 1) * ah
    return np.stack([cy - (h - 1) / 2, cx - (w - 1) / 2, cy + (h - 1) / 2, cx + (w - 1) / 2], axis=-1)


def decode_to_centroids_gt(offsets, anchors_centroids, stds):
    """DataEncoderDecoder.decode_to_centroids (datacoder.py:368-388): ground-truth offsets (A,4) -> (A,4)."""
    dt = offsets.dtype
    acx, acy, aw, ah = (anchors_centroids[:, i].astype(dt) for i in range(4))
    sx, sy, sw, sh = (np.asarray(s, dt) for s in stds)
    nb = (np.abs(offsets).sum(axis=-1) > 0).astype(dt)
    cx = (offsets[:, 0] * sx * aw + acx) * nb
    cy = (offsets[:, 1] * sy * ah + acy) * nb
    w = (np.exp(offsets[:, 2] * sw) - 1) * aw * nb
    h = (np.exp(offsets[:, 3] * sh) - 1) * ah * nb
    return np.stack([cx, cy, w, h], axis=1)


def _iou_tf(a, b):
    """IoU used by TF's NMS kernels: raw corner coords (y1,x1,y2,x2), no +1, area<=0 -> 0 (App. B.9)."""
    f = np.float32
    ya0, xa0, ya1, xa1 = min(a[0], a[2]), min(a[1], a[3]), max(a[0], a[2]), max(a[1], a[3])
    yb0, xb0, yb1, xb1 = min(b[0], b[2]), min(b[1], b[3]), max(b[0], b[2]), max(b[1], b[3])
    area_a = f(ya1 - ya0) * f(xa1 - xa0)
    area_b = f(yb1 - yb0) * f(xb1 - xb0)
    if area_a <= 0 or area_b <= 0:
        return f(0)
    iy0, ix0, iy1, ix1 = max(ya0, yb0), max(xa0, xb0), min(ya1, yb1), min(xa1, xb1)
    inter = f(max(f(iy1 - iy0), f(0))) * f(max(f(ix1 - ix0), f(0)))
    return f(inter / f(f(area_a + area_b) - inter))


def combined_nms(corners, probs, max_per_class, max_total, iou_thr, score_thr):
    """layers.NonMaximumSuppression.call (layers.py:141-162) == tf.image.combined_non_max_suppression
    (boxes shared across classes, background class competes, pad_per_class=False, clip_boxes=False) + repack.
    corners (B,A,4) ymin,xmin,ymax,xmax; probs (B,A,C) -> out (B,max_total,6) = label,prob,xmin,ymin,xmax,ymax; valid (B,).
    Equal scores are ordered by (score desc, anchor index asc, class asc) -- TF leaves this unspecified."""
    f = np.float32
    b, a, c = probs.shape
    out = np.zeros((b, max_total, 6), f)
    valid = np.zeros(b, np.int32)
    for bi in range(b):
        picked = []  # (score, anchor, class)
        for cl in range(c):
            sc = probs[bi, :, cl]
            cand = np.nonzero(sc > f(score_thr))[0]
            cand = cand[np.lexsort((cand, -sc[cand].astype(np.float64)))]
            kept = []
            for i in cand:
                if len(kept) >= max_per_class:
                    break
                if all(_iou_tf(corners[bi, i], corners[bi, j]) <= f(iou_thr) for j in kept):
                    kept.append(int(i))
            picked += [(float(sc[i]), i, cl) for i in kept]
        picked.sort(key=lambda t: (-t[0], t[1], t[2]))
        picked = picked[:max_total]
        valid[bi] = len(picked)
        for r, (s, i, cl) in enumerate(picked):
            y0, x0, y1, x1 = corners[bi, i]
            out[bi, r] = (cl, s, x0, y0, x1, y1)
    return out, valid


def seg_suppress(mask_prob, probs):
    """layers.SegmentationSuppression.call (layers.py:203-210): class present anywhere in the BATCH (quirk Q6)."""
    cls = mask_prob.argmax(axis=-1)
    present = np.zeros(mask_prob.shape[-1], probs.dtype)
    present[np.unique(cls)] = 1.0
    return probs * present


# --------------------------------------------------------------------------------------------- optimizer
def adam_step(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-7):
    """Keras 2.13 Adam (App. B.10) -> (p, m, v)."""
    m = m + (g - m) * (1 - b1)
    v = v + (g * g - v) * (1 - b2)
    alpha = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    p = p - alpha * m / (np.sqrt(v) + eps)
    return p.astype(g.dtype), m.astype(g.dtype), v.astype(g.dtype)


# --------------------------------------------------------------------------------------------- training metrics
KEPS = 1e-7   # tf.keras.backend.epsilon()


def metric_mask_iou(y_true, y_pred, class_weights):
    """metrics.jaccard_iou_segmentation_masks_metric (metrics.py:35-47): SOFT Jaccard on probabilities -> (batch,)."""
    y_true, y_pred = np.asarray(y_true, np.float64), np.asarray(y_pred, np.float64)
    inter = (y_true * y_pred).sum(axis=(1, 2))
    total = (y_true + y_pred).sum(axis=(1, 2))
    return (inter / (total - inter + KEPS) * np.asarray(class_weights, np.float64)[None]).sum(-1)


def metric_label_accuracy(y_true, y_pred, class_weights):
    """metrics.categorical_accuracy_metric (metrics.py:204-216): equal(one_hot(argmax p), y_true) counted per class over the
    boxes (agreeing zeros count), / #boxes, weighted sum -> (batch,)."""
    y_true, y_pred = np.asarray(y_true, np.float32), np.asarray(y_pred, np.float32)
    onehot = np.eye(y_pred.shape[-1], dtype=np.float32)[y_pred.argmax(-1)]
    tp = (onehot == y_true).astype(np.float64).sum(axis=1)
    return (tp / y_true.shape[1] * np.asarray(class_weights, np.float64)[None]).sum(-1)


def metric_box_iou(y_true, y_pred, cx, cy, w, h, stds):
    """metrics.jaccard_iou_bounding_boxes_metric (metrics.py:76-171), every convention of the reference kept: not_background
    from |y_true|, widths clamped at 0, corners c -+ (w-1)/2, areas w*h, +1 extents of the intersection, epsilon, and
    sum / #non-background (NaN without objects) -> (batch,)."""
    y_true, y_pred = np.asarray(y_true, np.float64), np.asarray(y_pred, np.float64)
    cx, cy, w, h = (np.asarray(v, np.float64) for v in (cx, cy, w, h))
    nb = (np.abs(y_true).sum(-1) > 0).astype(np.float64)

    def dec(o):
        x = (o[..., 0] * stds[0] * w + cx) * nb
        y = (o[..., 1] * stds[1] * h + cy) * nb
        ww = np.maximum(0.0, (np.exp(o[..., 2] * stds[2]) - 1.0) * w) * nb
        hh = np.maximum(0.0, (np.exp(o[..., 3] * stds[3]) - 1.0) * h) * nb
        return (x - (ww - 1) / 2) * nb, (y - (hh - 1) / 2) * nb, (x + (ww - 1) / 2) * nb, (y + (hh - 1) / 2) * nb, ww, hh

    px0, py0, px1, py1, pw, ph = dec(y_pred)
    tx0, ty0, tx1, ty1, tw, th = dec(y_true)
    wi = np.maximum(0.0, np.minimum(tx1, px1) - np.maximum(tx0, px0) + 1.0) * nb
    hi = np.maximum(0.0, np.minimum(ty1, py1) - np.maximum(ty0, py0) + 1.0) * nb
    inter = wi * hi
    with np.errstate(invalid="ignore", divide="ignore"):
        return (inter / (pw * ph + tw * th - inter + KEPS)).sum(-1) / nb.sum(-1)


# ---------------------------------------------------------------------------------------------- input pipeline (datacoder.py:302-347)
def expand_inputs(images_u8, mask_index_u8, flip, num_classes):
    """read_and_encode's tensor part on a batch: tf.cast(image, float32) (:327), tf.one_hot(mask, depth) (:332: out-of-range index ->
    all-zero row), tf.image.flip_left_right of both where flip[n] (:341-342)."""
    img = np.asarray(images_u8).astype(np.float32)
    idx = np.asarray(mask_index_u8).astype(np.int64)
    onehot = (idx[..., None] == np.arange(num_classes)).astype(np.float32)
    if flip is not None:
        f = np.asarray(flip).astype(bool)
        img[f] = img[f][:, :, ::-1]
        onehot[f] = onehot[f][:, :, ::-1]
    return img, onehot


def flip_gt_boxes(gt, image_width):
    """horizontal flip of (label, xmin, ymin, xmax, ymax) rows: x -> W - x with W the image WIDTH, not W - 1 (datacoder.py:202-203)"""
    g = np.asarray(gt, np.float32).reshape(-1, 5)
    out = g.copy()
    out[:, 1] = np.float32(image_width) - g[:, 3]
    out[:, 3] = np.float32(image_width) - g[:, 1]
    return out
