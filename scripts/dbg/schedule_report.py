"""which ops Engine._schedule puts in the trunk / detection / mask phases (debug aid; needs a GPU for the engine's buffers)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))
import bench
from ssdseglib import _hip as H
ctx = H.Context(0)
for w in sys.argv[1:] or ["full", "shufflenet-q1fixed"]:
    step = bench.STEPS[w](ctx, 2, 0, None)
    trunk, det, mask, jb = step.eng._schedule()
    print(w, "trunk", len(trunk), "det", len(det), "mask", len(mask), "join_before", len(jb))
    if det:
        print("   first det ops:", [op.name for op in det[:4]], " kept with the trunk:", [op.name for op in trunk if op.reach and op.reach <= step.eng.DET_OUTPUTS])
