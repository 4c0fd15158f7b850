"""which parameter gradients differ between SSDSEG_DET_SIDE=1 and 0 (debug aid for Engine._schedule)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd"))
import numpy as np
import ssdseglib
from ssdseglib import _engine as E, _hip as H
from tests.test_gpu_full_model import build, make_targets, CW, SHAPE
ctx = H.Context(0)
rng = np.random.default_rng(5)
batch = 3
boxes, builder, model = build(seed=23)
enc, gts, targets = make_targets(rng, boxes, batch)
model.compile(optimizer=ssdseglib.optimizers.Adam(learning_rate=1e-3),
              loss={'output-mask': ssdseglib.losses.cross_entropy(classes_weights=CW), 'output-labels': ssdseglib.losses.confidence_loss,
                    'output-boxes': ssdseglib.losses.localization_loss})
eng = E.Engine(model, batch, training=True, ctx=ctx)
eng.configure_losses(model._compiled["loss"], model._compiled["loss_weights"])
x = rng.integers(0, 256, (batch,) + SHAPE).astype(np.float32)
trunk, det, mask, jb = eng._schedule()
print(len(trunk), len(det), len(mask), [op.name for op in mask if id(op) in jb][:5])
order = [("det" if op in det else ("mask" if op in mask else "trunk")) for op in eng.ops]
runs = []
for o in order:
    if not runs or runs[-1][0] != o: runs.append([o, 0])
    runs[-1][1] += 1
print("op order in layer order:", runs)
w13 = [op.name for op in eng.ops if any(getattr(getattr(op, "inp", None), "store", None) is st for st in eng.stores if st.name == "backbone-block13-expand-conv")]
print("consumers of block13-expand in op order:", w13)
res = {}
taps = {st.name: st for st in eng.stores if any(k in st.name for k in ("block16-project-conv", "block13-expand-conv", "block3-expand-conv"))}
print(list(taps))
tg = {}
for mode in ("0", "1", "0", "1"):
    os.environ["SSDSEG_DET_SIDE"] = mode
    eng.set_input(x); eng.set_targets(targets)
    eng.forward(); eng.backward(); ctx.sync()
    g = {(l.name, w): eng.grad_view(l, w).download() for l in model.layers for w in l.trainable_names}
    if mode in res:
        same = all(np.array_equal(g[k], res[mode][k]) for k in g)
        print("mode", mode, "repeat identical:", same)
    res[mode] = g
    tg[mode] = {k: st.grad.download() for k, st in taps.items()}
for m in ("1",):
    bad = [k for k in res["0"] if not np.array_equal(res["0"][k], res[m][k])]
    print("mode", m, ":", len(bad), "of", len(res["0"]), "gradients differ;  tap gradients differ:",
          {k: float(np.abs(tg["0"][k] - tg[m][k]).max() / max(np.abs(tg["0"][k]).max(), 1e-30)) for k in tg["0"]})
    print("   first:", bad[:3], " last:", bad[-3:])
