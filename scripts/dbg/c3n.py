import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))
import numpy as np
from oracle import np_ops as O
from ssdseglib import _hip as H
from tests.test_gpu_head_ops import view_inputs, gview_inputs, rel_err
os.environ["SSDSEG_CONV3_NARROW"] = "1"
ctx = H.Context(0)
rng = np.random.default_rng(1993)
n, h, w, cin, cout = 1, 5, 5, 24, 8
act = O.ACT_RELU6
x, sc, sh, a = view_inputs(rng, (n, h, w, cin), act)
wgt = (rng.normal(0, 1, (3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
gv, dy = gview_inputs(rng, (n, h, w, cout), O.ACT_RELU6)
bufs = [ctx.array(v) for v in gv]
_, dw_ref, _ = O.conv2d_bwd(a.astype(np.float64), wgt.astype(np.float64), dy.astype(np.float64))
dx_, dsc, dsh = ctx.array(x), ctx.array(sc), ctx.array(sh)
ldi = cin + 8
xw = np.zeros((n, h, w, ldi), np.float32); xw[..., 4:4 + cin] = x
dxw = ctx.array(xw)
dmat = ctx.array(dy)
ddw = ctx.empty(wgt.shape)
for name, xv, ld, g in [("bn,dense", H.view(dx_, dsc, dsh, act), cin, H.gview(*bufs, act=O.ACT_RELU6)),
                        ("id,dense", H.view(dx_, dsc, dsh, act), cin, H.gview(dmat)),
                        ("bn,slice", H.view(dxw.view(4, (dxw.size - 4,)), dsc, dsh, act), ldi, H.gview(*bufs, act=O.ACT_RELU6)),
                        ("id,slice", H.view(dxw.view(4, (dxw.size - 4,)), dsc, dsh, act), ldi, H.gview(dmat))]:
    ddw.upload(np.zeros(wgt.shape, np.float32))
    ctx.call("ssdseg_conv3x3_bwd_weight", xv, ld, g, ddw, n, h, w, cin, cout)
    print(name, rel_err(ddw.download(), dw_ref))
