"""time the fused expand-conv backward (ssdseg_pwconv_bwd) at the block-1/2/3 shapes, with its own forward output as the view's y"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))
import numpy as np
from oracle import np_ops as O
from ssdseglib import _hip as H
ctx = H.Context(0)
rng = np.random.default_rng(3)
for (m, k, n) in ((2457600, 16, 96), (614400, 24, 144)):
    x = rng.standard_normal((m, k), dtype=np.float32); sc = rng.uniform(0.5, 1.5, k).astype(np.float32); sh = rng.uniform(-1, 3, k).astype(np.float32)
    w = (rng.normal(0, 1, (k, n)) / np.sqrt(k)).astype(np.float32)
    dx_, dsc, dsh, dw_ = ctx.array(x), ctx.array(sc), ctx.array(sh), ctx.array(w)
    y = ctx.empty((m, n)); stats = ctx.empty((ctx.parts("ssdseg_pwconv_parts", m, n), 2, n))
    ctx.call("ssdseg_pwconv_fwd", H.view(dx_, dsc, dsh, O.ACT_RELU6), k, dw_, y, n, m, k, n, stats)
    g = ctx.array(rng.standard_normal((m, n), dtype=np.float32))
    co = [ctx.array(v) for v in (rng.uniform(0.5, 1.5, n).astype(np.float32), rng.uniform(-1, 3, n).astype(np.float32),
                                 rng.normal(0, 0.1, n).astype(np.float32), rng.normal(0, 0.1, n).astype(np.float32))]
    gv = H.gview(g, y, *co, act=O.ACT_RELU6)
    ddx, ddw = ctx.empty((m, k)), ctx.empty((k, n))
    for mode in ("1", "0"):
        os.environ["SSDSEG_WRES_RC"] = mode
        for _ in range(2):
            ctx.call("ssdseg_pwconv_bwd", H.view(dx_, dsc, dsh, O.ACT_RELU6), k, gv, n, dw_, ddx, k, ddw, m, k, n, None, 0, 0)
        ctx.sync(); ctx.timing(True); ctx.timing_reset()
        for _ in range(5):
            ctx.call("ssdseg_pwconv_bwd", H.view(dx_, dsc, dsh, O.ACT_RELU6), k, gv, n, dw_, ddx, k, ddw, m, k, n, None, 0, 0)
        ctx.sync()
        for name, r in ctx.timing_report().items():
            if r["count"] and "wres" in name:
                print(f"{m}x{k}x{n} RC={mode} {name:34s} {r['ms'] / r['count'] * 1e3:8.1f} us")
        ctx.timing(False)
