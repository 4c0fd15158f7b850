import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))
import numpy as np
from oracle import np_ops as O
from ssdseglib import _hip as H
from tests.test_gpu_head_ops import view_inputs, gview_inputs, rel_err
ctx = H.Context(0)
rng = np.random.default_rng(1993)
m, k, n = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (70001, 24, 144)
act = O.ACT_RELU6
x, sc, sh, a = view_inputs(rng, (m, k), act)
bufs, dy = gview_inputs(rng, (m, n), act)
dx_, dsc, dsh = ctx.array(x), ctx.array(sc), ctx.array(sh)
dw_ref = a.astype(np.float64).T @ dy.astype(np.float64)
for name, gv, ref in [("identity", H.gview(ctx.array(dy)), dw_ref), ("bn", H.gview(*[ctx.array(v) for v in bufs], act=act), dw_ref)]:
    dwg = ctx.empty((k, n))
    ctx.call("ssdseg_pwconv_bwd_weight", H.view(dx_, dsc, dsh, act), k, gv, n, dwg, m, k, n)
    got = dwg.download()
    err = np.abs(got - ref) / np.abs(ref).max()
    bad = np.argwhere(err > 1e-4)
    print(name, "rel_err", err.max(), "bad", len(bad), "of", k * n)
    if len(bad):
        print("  bad k:", sorted(set(bad[:, 0].tolist()))[:40])
        print("  bad n:", sorted(set(bad[:, 1].tolist()))[:60])
        i, j = bad[0]
        print("  first", i, j, got[i, j], ref[i, j], got[i, j] - ref[i, j])
