#!/usr/bin/env python3
"""Decoder 3x3 conv (304 -> 256 at 120x160, batch 32: the layer that is 45% of the full step) forward / input gradient / weight
gradient in isolation, under the kernel switches of the environment.  usage: python scripts/conv3_decoder_time.py [reps] [batch] [cin] [cout]
prints per kernel: launches, ms per launch, TFLOP/s by the flops the kernel executes (Winograd kernels: 16/36 of the direct
convolution's; the direct-equivalent figure beside it); and a parity check against the direct kernels."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))
import numpy as np
from ssdseglib import _hip as H

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cin = int(sys.argv[3]) if len(sys.argv) > 3 else 304
cout = int(sys.argv[4]) if len(sys.argv) > 4 else 256
h, w, ldx = 120, 160, cin
ctx = H.Context(0)
rng = np.random.default_rng(7)
x = ctx.array(np.clip(rng.normal(0.5, 1.5, (n, h, w, cin)), 0, 6).astype(np.float32))
wgt = ctx.array((rng.normal(0, 1, (3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32))
dy = ctx.array(rng.normal(0, 1, (n, h, w, cout)).astype(np.float32))
y, dx, dw = ctx.empty((n, h, w, cout)), ctx.empty((n, h, w, cin)), ctx.empty((3, 3, cin, cout))
nparts = ctx.parts("ssdseg_conv3x3_parts", n, h, w, cin, cout)
_keep = os.environ.get("SSDSEG_CONV3_WINOGRAD")
os.environ["SSDSEG_CONV3_WINOGRAD"] = "0"      # (the direct kernels of the parity check write one row per 8 x 32 tile: size for both)
nparts = max(nparts, ctx.parts("ssdseg_conv3x3_parts", n, h, w, cin, cout))
if _keep is None:
    del os.environ["SSDSEG_CONV3_WINOGRAD"]
else:
    os.environ["SSDSEG_CONV3_WINOGRAD"] = _keep
stats = ctx.zeros((nparts, 2, cout))


def run():
    ctx.call("ssdseg_conv3x3_fwd", H.view(x), ldx, wgt, y, n, h, w, cin, cout, stats)
    ctx.call("ssdseg_conv3x3_bwd_data", H.gview(dy), wgt, dx, ldx, n, h, w, cin, cout, 0)
    ctx.call("ssdseg_conv3x3_bwd_weight", H.view(x), ldx, H.gview(dy), dw, n, h, w, cin, cout)


run(); ctx.sync()
ctx.timing(True)
for _ in range(reps):
    run()
ctx.sync()
for name, r in ctx.timing_report().items():
    if r["count"]:
        ms = r["ms"] / r["count"]
        tf = r['flops'] / r['count'] / ms / 1e9 if ms else 0
        print(f"{name[:60]:60s} x{r['count']:3d}  {ms:8.3f} ms/launch  {tf:7.1f} TF" + (f"  ({tf * (4.0 if 'wino4' in name else 2.25):6.1f} TF direct-equivalent)" if "wino" in name and tf else ""))
ctx.timing(False)
got = {"y": y.download(), "dx": dx.download(), "dw": dw.download(), "stats": stats.download().sum(0)}
if os.environ.get("SSDSEG_CONV3_WINOGRAD", "") != "0":
    os.environ["SSDSEG_CONV3_WINOGRAD"] = "0"
    stats.upload(np.zeros((nparts, 2, cout), np.float32))
    run(); ctx.sync()
    ref = {"y": y.download(), "dx": dx.download(), "dw": dw.download(), "stats": stats.download().sum(0)}
    for k in got:
        print(f"{k}: max |winograd - direct| / max |direct| = {np.abs(got[k] - ref[k]).max() / np.abs(ref[k]).max():.2e}")
