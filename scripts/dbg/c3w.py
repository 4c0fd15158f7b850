import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))
import numpy as np
from oracle import np_ops as O
from ssdseglib import _hip as H
from tests.test_gpu_head_ops import view_inputs, gview_inputs, rel_err
ctx = H.Context(0)
rng = np.random.default_rng(1993)
act = O.ACT_RELU6
for (n, h, w, cin, cout) in [(1, 2, 32, 32, 32), (1, 5, 5, 24, 8), (1, 4, 32, 64, 128), (2, 9, 40, 72, 96), (1, 12, 16, 304, 256)]:
    x, sc, sh, a = view_inputs(rng, (n, h, w, cin), act)
    wgt = (rng.normal(0, 1, (3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    gv, dy = gview_inputs(rng, (n, h, w, cout), O.ACT_RELU6)
    _, dw_ref, _ = O.conv2d_bwd(a.astype(np.float64), wgt.astype(np.float64), dy.astype(np.float64))
    dx_, dsc, dsh = ctx.array(x), ctx.array(sc), ctx.array(sh)
    dmat = ctx.array(dy)
    ddw = ctx.zeros(wgt.shape)
    ctx.call("ssdseg_conv3x3_bwd_weight", H.view(dx_, dsc, dsh, act), cin, H.gview(dmat), ddw, n, h, w, cin, cout)
    got = ddw.download()
    err = np.abs(got - dw_ref).max(axis=(2, 3)) / np.abs(dw_ref).max()
    print((n, h, w, cin, cout), "per-tap rel err:\n", np.array2string(err, precision=2))
    kerr = np.abs(got - dw_ref).max(axis=(0, 1, 3)) / np.abs(dw_ref).max()
    nerr = np.abs(got - dw_ref).max(axis=(0, 1, 2)) / np.abs(dw_ref).max()
    print("  bad k:", np.nonzero(kerr > 1e-4)[0][:20], " bad n:", np.nonzero(nerr > 1e-4)[0][:20])
