"""Pins for the NumPy oracle itself (CPU only).

The reference ships no tests and TensorFlow is not installable here, so TF-level parity of conv/BN/resize is
UNPINNED; what this file does instead:
  * cross-checks every float op of np_ops against torch-CPU (independent implementation + autograd);
  * hand-worked known answers for the matching rule, mining (B.8) and NMS (B.9);
  * algebraic identities the reference's code implies (encode <-> decode, zero-loss cases).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import np_ops as O


def t_nhwc(x):
    return torch.tensor(x).permute(0, 3, 1, 2)


def pad_same(xt, h, w, k, s, d):
    _, pt, pb = O.same_pad(h, k, s, d)
    _, pl, pr = O.same_pad(w, k, s, d)
    return F.pad(xt, (pl, pr, pt, pb))


def test_same_padding_rules():
    # even sizes with stride 2 pad (0,1); 15 -> 8 pads (1,1); atrous d pads (d,d)
    assert O.same_pad(480, 3, 2) == (240, 0, 1)
    assert O.same_pad(15, 3, 2) == (8, 1, 1)
    assert O.same_pad(20, 3, 2) == (10, 0, 1)
    assert O.same_pad(30, 3, 1, 12) == (30, 12, 12)


@pytest.mark.parametrize("s,d", [(1, 1), (2, 1), (1, 3)])
def test_conv_and_dwconv_vs_torch(s, d):
    rng = np.random.default_rng(0)
    x = rng.normal(size=(2, 9, 7, 8)).astype(np.float64)
    w = rng.normal(size=(3, 3, 8, 5)).astype(np.float64)
    wd = rng.normal(size=(3, 3, 8)).astype(np.float64)
    xt = t_nhwc(x).requires_grad_(True)
    wt = torch.tensor(w).permute(3, 2, 0, 1).requires_grad_(True)
    y = F.conv2d(pad_same(xt, 9, 7, 3, s, d), wt, stride=s, dilation=d)
    assert np.allclose(O.conv2d_fwd(x, w, s, d), y.detach().permute(0, 2, 3, 1).numpy(), atol=1e-10)
    g = rng.normal(size=tuple(y.permute(0, 2, 3, 1).shape))
    y.backward(t_nhwc(g))
    dx, dw, db = O.conv2d_bwd(x, w, g, s, d)
    assert np.allclose(dx, xt.grad.permute(0, 2, 3, 1).numpy(), atol=1e-10)
    assert np.allclose(dw, wt.grad.permute(2, 3, 1, 0).numpy(), atol=1e-10)
    assert np.allclose(db, g.sum((0, 1, 2)))
    xt2 = t_nhwc(x).requires_grad_(True)
    wdt = torch.tensor(wd).permute(2, 0, 1)[:, None].clone().requires_grad_(True)
    y2 = F.conv2d(pad_same(xt2, 9, 7, 3, s, d), wdt, stride=s, dilation=d, groups=8)
    assert np.allclose(O.dwconv_fwd(x, wd, s, d), y2.detach().permute(0, 2, 3, 1).numpy(), atol=1e-10)
    g2 = rng.normal(size=tuple(y2.permute(0, 2, 3, 1).shape))
    y2.backward(t_nhwc(g2))
    dx2, dw2 = O.dwconv_bwd(x, wd, g2, s, d)
    assert np.allclose(dx2, xt2.grad.permute(0, 2, 3, 1).numpy(), atol=1e-10)
    assert np.allclose(dw2, wdt.grad[:, 0].permute(1, 2, 0).numpy(), atol=1e-10)


def test_batchnorm_train_vs_torch():
    rng = np.random.default_rng(1)
    y = rng.normal(0.5, 2, (3, 5, 4, 6))
    gamma, beta = rng.uniform(0.5, 1.5, 6), rng.normal(0, 1, 6)
    yt = t_nhwc(y).requires_grad_(True)
    gt, bt = torch.tensor(gamma, requires_grad=True), torch.tensor(beta, requires_grad=True)
    rm, rv = torch.zeros(6, dtype=torch.float64), torch.ones(6, dtype=torch.float64)
    z = F.batch_norm(yt, rm, rv, gt, bt, training=True, momentum=0.01, eps=1e-3)     # torch momentum = 1 - keras momentum
    zo, cache = O.bn_train_fwd(y, gamma, beta, 1e-3)
    assert np.allclose(zo, z.detach().permute(0, 2, 3, 1).numpy(), atol=1e-10)
    mm, mv = O.bn_moving_update(np.zeros(6), np.ones(6), cache, 0.99)
    assert np.allclose(mm, rm.numpy()) and np.allclose(mv, rv.numpy())                # Bessel-corrected moving variance (App. B.3)
    g = rng.normal(size=y.shape)
    z.backward(t_nhwc(g))
    dy, dg, db = O.bn_train_bwd(g, y, gamma, cache)
    assert np.allclose(dy, yt.grad.permute(0, 2, 3, 1).numpy(), atol=1e-10)
    assert np.allclose(dg, gt.grad.numpy()) and np.allclose(db, bt.grad.numpy())


@pytest.mark.parametrize("h,w,fy,fx", [(6, 8, 4, 4), (1, 1, 30, 40), (5, 7, 2, 8)])
def test_bilinear_vs_torch(h, w, fy, fx):
    rng = np.random.default_rng(2)
    x = rng.normal(size=(2, h, w, 3))
    xt = t_nhwc(x).requires_grad_(True)
    y = F.interpolate(xt, scale_factor=(fy, fx), mode="bilinear", align_corners=False)   # half-pixel centres (App. B.5)
    assert np.allclose(O.bilinear_fwd(x, fy, fx), y.detach().permute(0, 2, 3, 1).numpy(), atol=1e-10)
    g = rng.normal(size=(2, h * fy, w * fx, 3))
    y.backward(t_nhwc(g))
    assert np.allclose(O.bilinear_bwd(g, fy, fx), xt.grad.permute(0, 2, 3, 1).numpy(), atol=1e-10)


def test_relu6_gap_softmax_losses_vs_torch():
    rng = np.random.default_rng(3)
    z = np.array([-1.0, 0.0, 3.0, 6.0, 7.5])
    assert np.array_equal(O.act_fwd(z, O.ACT_RELU6), [0, 0, 3, 6, 6])
    assert np.array_equal(O.act_mask(z, O.ACT_RELU6), [0, 0, 1, 0, 0])          # strict 0 < z < 6 (App. B.4)
    assert np.array_equal(O.act_fwd(z, O.ACT_ZERO), np.zeros(5))                 # quirk Q1
    x = rng.normal(size=(2, 3, 4, 5))
    assert np.allclose(O.gap_fwd(x)[:, 0, 0], x.mean((1, 2)))
    logits = rng.normal(size=(2, 6, 5, 4))
    lt = torch.tensor(logits, requires_grad=True)
    y = np.eye(4)[rng.integers(0, 4, (2, 6, 5))]
    w = np.array([0.05, 0.575, 0.135, 0.24])
    p = torch.softmax(lt, -1)
    loss_t = -(torch.tensor(y) * torch.log(torch.clamp(p, 1e-7, 1 - 1e-7))).sum((1, 2)) @ torch.tensor(w)
    loss_t.sum().backward()
    pn = O.softmax(logits)
    loss, dp = O.cross_entropy_loss(y, pn, w)
    assert np.allclose(loss, loss_t.detach().numpy())
    assert np.allclose(O.softmax_bwd(pn, dp), lt.grad.numpy(), atol=1e-10)
    yb = rng.normal(size=(2, 7, 4)) * (rng.uniform(size=(2, 7, 1)) < 0.5)
    pb = rng.normal(size=(2, 7, 4)) * 2
    pbt = torch.tensor(pb, requires_grad=True)
    nb = (torch.tensor(yb).abs().sum(-1) > 0).double()
    sl1 = F.smooth_l1_loss(pbt, torch.tensor(yb), reduction="none", beta=1.0).sum(-1) * nb
    lt2 = sl1.sum(-1) / torch.clamp(nb.sum(-1), min=1.0)
    lt2.sum().backward()
    l2, d2 = O.localization_loss(yb, pb)
    assert np.allclose(l2, lt2.detach().numpy()) and np.allclose(d2, pbt.grad.numpy())


def test_adam_vs_torch_first_steps():
    rng = np.random.default_rng(4)
    p0, g1, g2 = rng.normal(size=50), rng.normal(size=50), rng.normal(size=50)
    pt = torch.tensor(p0, requires_grad=True)
    opt = torch.optim.Adam([pt], lr=1e-2, betas=(0.9, 0.999), eps=1e-7)
    p, m, v = p0.copy(), np.zeros(50), np.zeros(50)
    for step, g in enumerate((g1, g2), start=1):
        pt.grad = torch.tensor(g)
        # Keras adds epsilon to sqrt(v) (App. B.10), torch to sqrt(v_hat): identical iff eps_torch = eps / sqrt(1 - b2^t)
        opt.param_groups[0]["eps"] = 1e-7 / np.sqrt(1 - 0.999 ** step)
        opt.step()
        p, m, v = O.adam_step(p, g, m, v, step, lr=1e-2)
    assert np.allclose(p, pt.detach().numpy(), rtol=0, atol=1e-12)


# ---------------------------------------------------------------------------- index semantics: hand-worked vectors
def test_encode_matching_rule_hand_worked():
    """three anchors, two ground-truth boxes; App. B.7: rows = step-1 rows (g ascending) ++ step-2 rows (d ascending),
    first occurrences kept, sequential scatter -> last row of an anchor wins."""
    anchors = np.array([[0, 0, 9, 9], [20, 0, 29, 9], [40, 0, 49, 9]], np.float32)
    gt = np.array([[1, 0, 0, 9, 9],        # g0 == anchor 0 (IoU 1)
                   [2, 1, 0, 10, 9],       # g1 overlaps anchor 0 strongly (IoU 9*10/(100+100-90) = .818) and nothing else
                   [3, 41, 0, 49, 4]],     # g2: best anchor 2 with IoU 45/100 = .45 < threshold
                  np.float32)
    labels, boxes, match = O.encode_targets(anchors, gt, 4, 0.5, (0.1, 0.1, 0.2, 0.2))
    # anchor 0: S1 = {g0, g1} (both pick anchor 0), s2 = g0 (IoU 1 > .5) is in S1 -> last step-1 row wins: g1
    assert match.tolist() == [1, -1, 2]
    assert labels[0].tolist() == [0, 0, 1, 0] and labels[1].tolist() == [1, 0, 0, 0] and labels[2].tolist() == [0, 0, 0, 1]
    assert np.all(boxes[1] == 0)
    # offsets use log(gw/aw + 1) (non-standard +1, datacoder.py:268): g1 on anchor 0 -> widths 10 vs 10
    assert np.isclose(boxes[0, 2], np.log(10 / 10 + 1) / 0.2) and np.isclose(boxes[0, 0], (5.5 - 4.5) / 10 / 0.1)
    # no ground truth -> all background
    l0, b0, m0 = O.encode_targets(anchors, np.zeros((0, 5), np.float32), 4, 0.5, (0.1, 0.1, 0.2, 0.2))
    assert (l0[:, 0] == 1).all() and (b0 == 0).all() and (m0 == -1).all()


def test_encode_decode_round_trip(golden_dir):
    d = np.load(f"{golden_dir}/anchors_nb03.npz")
    rng = np.random.default_rng(5)
    gt = np.array([[1, 100, 120, 260, 300], [3, 400, 50, 520, 400], [2, 10, 10, 60, 90]], np.float32)
    stds = (0.1, 0.1, 0.2, 0.2)
    labels, offsets, match = O.encode_targets(d["corners"], gt, 4, 0.525, stds)
    assert (match >= 0).sum() >= 3
    dec = O.decode_to_centroids_gt(offsets, d["centroids"], stds)
    pos = np.nonzero(match >= 0)[0]
    g = gt[match[pos]]
    want = np.stack([(g[:, 3] + g[:, 1]) / 2, (g[:, 4] + g[:, 2]) / 2, g[:, 3] - g[:, 1] + 1, g[:, 4] - g[:, 2] + 1], 1)
    assert np.abs(dec[pos] - want).max() < 2e-2            # exp(log(r + 1)) - 1 == r  (datacoder.py:268 <-> :373)
    assert np.all(dec[match < 0] == 0)


def test_topk_and_mining_hand_worked():
    v = np.array([0.5, 2.0, 2.0, 0.1, 2.0, 0.0], np.float32)
    assert O.topk_mask(v, 2).tolist() == [0, 1, 1, 0, 0, 0]        # ties: lower index first (App. B.8)
    assert O.topk_mask(v, 0).sum() == 0 and O.topk_mask(v, 6).sum() == 6
    # one positive anchor -> k = 3 hardest negatives of the WHOLE batch (batch-global pool, losses.py:113,127)
    y = np.zeros((2, 4, 4), np.float32); y[..., 0] = 1
    y[0, 0] = [0, 1, 0, 0]
    p = np.full((2, 4, 4), 0.25, np.float32)
    p[1, :, 0] = [0.1, 0.2, 0.3, 0.9]; p[1, :, 1] = 1 - p[1, :, 0]; p[1, :, 2:] = 0
    loss, dp, keep = O.confidence_loss(y, p)
    # background losses: image 0 -> 1.386 x3 (anchors 1..3), image 1 -> 2.303, 1.609, 1.204, 0.105; the three hardest of the
    # batch are 2.303, 1.609 and the FIRST of the tied 1.386s (lower flat index wins)
    assert keep.reshape(2, 4).tolist() == [[0, 1, 0, 0], [1, 1, 0, 0]]
    assert np.isclose(loss[0], -2 * np.log(0.25)) and np.isclose(loss[1], -(np.log(0.1) + np.log(0.2)))   # /max(#pos, 1)
    # no positives anywhere -> k = 0 -> zero loss (losses.py:113,170)
    y[0, 0] = [1, 0, 0, 0]
    assert np.all(O.confidence_loss(y, p)[0] == 0)
    assert np.all(O.localization_loss(np.zeros((2, 4, 4), np.float32), p)[0] == 0)    # losses.py:47


def test_combined_nms_hand_worked():
    # boxes as (ymin, xmin, ymax, xmax); A and B overlap heavily, C is apart
    corners = np.array([[[0, 0, 10, 10], [0, 1, 10, 11], [0, 30, 10, 40]]], np.float32)
    probs = np.array([[[0.1, 0.9, 0.0], [0.2, 0.8, 0.0], [0.3, 0.7, 0.0]]], np.float32)
    out, valid = O.combined_nms(corners, probs, max_per_class=2, max_total=4, iou_thr=0.5, score_thr=0.25)
    # class 1: A (0.9) kept, B suppressed by A (IoU 90/110), C (0.7) kept; class 0 (background competes, quirk Q7): C (0.3)
    assert valid.tolist() == [3]
    assert out[0, :, 0].tolist() == [1, 1, 0, 0] and np.allclose(out[0, :3, 1], [0.9, 0.7, 0.3])
    assert out[0, 0, 2:].tolist() == [0, 0, 10, 10] and out[0, 1, 2:].tolist() == [30, 0, 40, 10]   # repacked to xmin,ymin,xmax,ymax
    assert np.all(out[0, 3] == 0)                                                               # zero padding -> label 0
    # degenerate (zero-area) boxes never suppress each other (App. B.9)
    z = np.zeros((1, 2, 4), np.float32)
    out2, valid2 = O.combined_nms(z, np.array([[[0.0, 0.9], [0.0, 0.8]]], np.float32), 2, 4, 0.1, 0.5)
    assert valid2.tolist() == [2]


def test_seg_suppress_is_batch_global():
    mask = np.zeros((2, 2, 2, 4), np.float32); mask[..., 0] = 1
    mask[1, 0, 0] = [0, 0, 0, 1]                      # class 3 appears in image 1 only
    probs = np.ones((2, 5, 4), np.float32)
    out = O.seg_suppress(mask, probs)
    assert out[0, 0].tolist() == [1, 0, 0, 1]         # ... and is enabled for image 0 too (quirk Q6)


def test_metrics_hand_worked():
    """the three training metrics of reference metrics.py on cases small enough to do by hand"""
    w = (0.1, 0.2, 0.3, 0.4)
    # masks: 1x1x2 image; pixel 0 true class 1, predicted (0.5, 0.5, 0, 0); pixel 1 true class 0, predicted (1, 0, 0, 0)
    yt = np.array([[[[0, 1, 0, 0], [1, 0, 0, 0]]]], np.float32)
    yp = np.array([[[[0.5, 0.5, 0, 0], [1, 0, 0, 0]]]], np.float32)
    # class 0: inter 1, total 1 + 1.5 -> 1/1.5; class 1: inter 0.5, total 1 + 0.5 -> 0.5/1.0; classes 2, 3: 0/(0+eps) = 0
    assert np.allclose(O.metric_mask_iou(yt, yp, w), 0.1 * (1 / 1.5) + 0.2 * 0.5, atol=1e-6)
    # labels: 3 boxes, truth classes (0, 2, 2), arg-max predictions (0, 2, 1)
    lt = np.eye(4, dtype=np.float32)[[0, 2, 2]][None]
    lp = np.array([[[0.7, 0.1, 0.1, 0.1], [0.1, 0.2, 0.6, 0.1], [0.1, 0.5, 0.3, 0.1]]], np.float32)
    # per class agreeing entries over the 3 boxes: class 0: 3, class 1: 2 (box 2 predicted 1, truth 0 there), class 2: 2, class 3: 3
    assert np.allclose(O.metric_label_accuracy(lt, lp, w), (0.1 * 3 + 0.2 * 2 + 0.3 * 2 + 0.4 * 3) / 3)
    # boxes: one default box (cx, cy, w, h) = (50, 40, 20, 10), stds 1; truth offsets (0, 0, ln 2, ln 2) -> 20 x 10 box;
    # prediction identical -> IoU = (19+1)(9+1) / (200 + 200 - 200) = 1; a second, background anchor contributes nothing
    cx, cy, aw, ah = (np.array(v, np.float32) for v in ([50, 7], [40, 7], [20, 5], [10, 5]))
    t = np.array([[[0, 0, np.log(2.0), np.log(2.0)], [0, 0, 0, 0]]], np.float32)
    assert np.allclose(O.metric_box_iou(t, t, cx, cy, aw, ah, (1, 1, 1, 1)), 1.0, atol=1e-6)
    p = t.copy(); p[0, 0, 0] = 0.5                      # prediction shifted right by half a box width (10 px)
    # intersection extent x: min(59.5, 69.5) - max(40.5, 50.5) + 1 = 10, y: 10 -> 100 / (200 + 200 - 100)
    assert np.allclose(O.metric_box_iou(t, p, cx, cy, aw, ah, (1, 1, 1, 1)), 100 / 300, atol=1e-6)
    assert np.isnan(O.metric_box_iou(np.zeros_like(t), p, cx, cy, aw, ah, (1, 1, 1, 1))[0])   # no objects: 0/0 like the reference



def test_config0_single_480x640_forward_through_the_oracle():
    """BASELINE.json configs[0] on the CPU path available here (the NumPy restatement; TensorFlow cannot be imported): one
    480x640 image, inference mode, shapes [(1,480,640,4), (1,9600,4), (1,9600,4)] and the inference tail (1,10,6)"""
    import bench
    from oracle.np_model import NpModel
    boxes, builder = bench.build_models(seed=1993)
    model = builder.get_model_for_training('deeplabv3plus', 'ssdlite', segmentation_dilation_rates=(3, 6, 12))
    x = np.random.default_rng(0).integers(0, 256, (1, 480, 640, 3)).astype(np.float32)
    p_mask, p_labels, p_boxes = NpModel(model, dtype=np.float32).forward(x, training=False)
    assert p_mask.shape == (1, 480, 640, 4) and p_labels.shape == (1, 9600, 4) and p_boxes.shape == (1, 9600, 4)
    assert np.abs(p_mask.sum(-1) - 1).max() < 1e-5 and np.abs(p_labels.sum(-1) - 1).max() < 1e-5
    assert p_boxes.min() >= 0 and p_boxes.max() <= 6          # quirk Q3: offsets pass ReLU6
    inference = builder.get_model_for_inference(model_trained=model, max_number_of_boxes_per_class=4, max_number_of_boxes_per_sample=10,
                                                boxes_iou_threshold=0.5, labels_probability_threshold=0.2, suppress_background_boxes=False,
                                                use_segmentation_suppression=True)
    ref = NpModel(inference, dtype=np.float32)
    seg, det = ref.forward(x, training=False)
    assert seg.shape == (1, 480, 640, 4) and det.shape == (1, 10, 6)
