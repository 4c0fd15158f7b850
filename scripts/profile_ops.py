#!/usr/bin/env python3
"""Per-op (per-layer) kernel timing of one engine step via the C library's HIP-event registry, every kernel ALONE on the chip:
the weight-gradient side stream is disabled (ctx.side_enable(False)), so no row carries a co-running neighbour.  Each op runs REP
times; a row is one kernel symbol of one op: launches per op call, microseconds per op call (all its launches), and the rates of
the AVERAGE LAUNCH (algorithmic bytes / flops per launch over the launch's duration).
usage: python scripts/profile_ops.py [backbone|full|shufflenet] [batch]  -> table sorted by time (fwd and bwd separately)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))
import numpy as np
import bench
from ssdseglib import _hip as H

workload = sys.argv[1] if len(sys.argv) > 1 else "backbone"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx = H.Context(0)
step = bench.STEPS[workload](ctx, batch, 0, None)
for _ in range(2):
    step()
ctx.sync()
eng = step.eng
rows = []
ctx.side_enable(False)          # weight gradients on the main stream: nothing co-runs with the kernel being timed
ctx.timing(True)
REP = 3
for direction in ("fwd", "bwd"):
    ops = eng.ops if direction == "fwd" else list(reversed(eng.ops))
    if direction == "bwd":
        for s in eng.stores: s.gwritten = False
        if workload == "backbone":
            for i, g in enumerate(step.seeds): eng.seed_output_grad(i, g)
            eng.mark_output_grads_written()
    recs = [o.rec for o in eng.ops if hasattr(o, "rec")]            # BatchNorm records: `bwd_done` says a consumer already reduced them
    for op in ops:
        ctx.timing_reset()
        snap = [(s.gwritten, s.pending) for s in eng.stores]
        done = [r.bwd_done for r in recs]
        for r in range(REP):
            for s, (w, pend) in zip(eng.stores, snap): s.gwritten, s.pending = w, pend      # same accumulate flags / deferred residuals on every repeat
            for rec, d in zip(recs, done): rec.bwd_done = d        # ... and the same fused / separate BatchNorm-backward path
            getattr(op, direction)()
        rep = ctx.timing_report()
        for k, v in rep.items():
            if v["ms"] <= 0: continue
            assert v["count"] % REP == 0, (op.name, k, v["count"])     # every repeat launches the same kernels
            per_call = v["count"] // REP                                # launches of this symbol per op call
            ms_launch = v["ms"] / v["count"]                            # average launch
            rows.append((v["ms"] / REP, direction, op.name or type(op).__name__, k, v["bytes"] / v["count"], v["flops"] / v["count"], per_call, ms_launch))
ctx.timing(False)
ctx.side_enable(True)
rows.sort(reverse=True)
total = sum(r[0] for r in rows)
print(f"total kernel ms/step {total:.3f}")
print("   us/op-call dir op                                           kernel                                       launches  us/launch      GB/s     TF   (rates: per launch)")
for ms, d, name, k, b, f, cnt, msl in rows[:int(os.environ.get('TOP', '60'))]:
    print(f"{ms*1e3:9.1f} us {d} {name:44s} {k:44s} x{cnt} {msl*1e3:9.1f} {b/msl/1e6:8.0f} GB/s {f/msl/1e9:7.1f} TF")
