import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multi-task-learning-object-detection-semantic-segmentation_amd"))
import numpy as np
from tests.test_gpu_backbone import build_backbone, TAPS
from oracle.np_model import NpModel
from ssdseglib import _engine as E, _hip as H
batch, shape = int(sys.argv[1]), tuple(int(v) for v in sys.argv[2].split(","))
rng = np.random.default_rng(1993)
ctx = H.Context(0)
model = build_backbone(shape)
eng = E.Engine(model, batch, training=True, ctx=ctx)
x = rng.integers(0, 256, (batch,) + shape).astype(np.float32)
ref = NpModel(model, dtype=np.float64)
ref_out = ref.forward(x, training=True)
eng.set_input(x); eng.forward()
gouts = [rng.normal(0, 1, o.shape).astype(np.float32).astype(np.float64) for o in ref_out]
ref_grads = ref.backward(gouts)
for i, g in enumerate(gouts): eng.seed_output_grad(i, g)
eng.backward_from_outputs(); ctx.sync()
for l in model.layers:
    if not l.weights: continue
    scale = max(np.abs(ref_grads[l.name][w]).max() for w in l.trainable_names)
    for wname in l.trainable_names:
        got = eng.grad_view(l, wname).download()
        err = np.abs(got - ref_grads[l.name][wname]).max() / scale
        print(f"{l.name:45s} {wname:18s} {err:.2e} {scale:.2e}")
