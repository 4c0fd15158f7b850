#!/usr/bin/env python3
"""Stem conv forward (3 -> cout, stride 2, 480x640, batch 32) in isolation: direct kernel (csrc/stem.hip) per channel count, with / without
bias and BatchNorm statistics.  usage: python scripts/dbg/stem_time.py [reps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))
import numpy as np
from ssdseglib import _hip as H

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n, h, w = 32, 480, 640
ctx = H.Context(0)
rng = np.random.default_rng(3)
x = ctx.array(rng.integers(0, 256, (n, h, w, 3)).astype(np.float32))
for cout, bias, with_stats, scale, offset in [(32, False, True, 1 / 127.5, -1.0), (24, True, False, 1 / 127.5, -1.0), (24, False, True, 1 / 127.5, -1.0),
                                              (24, True, False, 1.0, 0.0), (16, False, True, 1 / 127.5, -1.0), (48, False, True, 1 / 127.5, -1.0), (64, False, True, 1 / 127.5, -1.0)]:
    wgt = ctx.array(rng.normal(0, 0.3, (3, 3, 3, cout)).astype(np.float32))
    b = ctx.array(rng.normal(0, 0.3, cout).astype(np.float32)) if bias else None
    y = ctx.empty((n, h // 2, w // 2, cout))
    stats = ctx.empty((ctx.parts("ssdseg_stem_conv_parts", n, h, w, cout), 2, cout)) if with_stats else None
    run = lambda: ctx.call("ssdseg_stem_conv_fwd", x, wgt, b, y, n, h, w, 3, cout, scale, offset, stats)
    run(); ctx.sync()
    ctx.timing(True); ctx.timing_reset()
    for _ in range(reps): run()
    ctx.sync()
    for name, r in ctx.timing_report().items():
        if r["count"]:
            ms = r["ms"] / r["count"]
            print(f"cout {cout:3d} bias {int(bias)} stats {int(with_stats)} scale {scale:.4f}: {name[:40]:40s} {ms * 1e3:8.1f} us  {r['bytes'] / r['count'] / ms / 1e6:7.0f} GB/s")
    ctx.timing(False)
