import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multi-task-learning-object-detection-semantic-segmentation_amd')
import bench
from ssdseglib import _hip as H
ctx = H.Context(0)
for wl in ("backbone", "full"):
    step = bench.STEPS[wl](ctx, 32, 0, None)
    for _ in range(3): step()
    ctx.sync()
    # host-only issue time: count ctypes calls and time them while the GPU is saturated (queue never drains)
    t0 = time.perf_counter(); 
    for _ in range(5): step()
    t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
    print(wl, "issue 5 steps: %.1f ms/step host, total %.1f ms/step" % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); ctx.sync()
    st = pstats.Stats(pr); st.sort_stats('cumulative')
    print("calls in one step:", st.total_calls)
