"""BASELINE.json configs[0]: ONE 480x640 image through the NB03 model in inference mode (`training=False`, reference
models.py:345-423 called as in NB03#cell31) -- MobileNetV2-SSDLite-DeepLabV3+ forward with moving statistics, then the inference
tail (segmentation suppression, box decode, combined per-class NMS).  The HIP path against the fp32/fp64 NumPy oracle of the same
graph and weights: shapes [(1,480,640,4), (1,10,6)], sum-of-softmax, probabilities / offsets / mask to 1e-3, and the index part
(NMS selection) exact when the oracle tail is fed the device's own head tensors."""
import numpy as np
import pytest

from oracle import np_ops as O
from oracle.np_model import NpModel

pytestmark = pytest.mark.gpu


def test_single_480x640_image_inference_forward_and_tail(ctx):
    import bench
    from ssdseglib import _engine as E, _graph as K
    E.set_default_context(ctx)
    boxes, builder = bench.build_models(seed=1993)
    model = builder.get_model_for_training('deeplabv3plus', 'ssdlite', segmentation_dilation_rates=(3, 6, 12))
    rng = np.random.default_rng(5)
    for l in model.layers:                  # a "trained" state: non-trivial BatchNorm parameters and moving statistics
        if type(l).__name__ == "BatchNormalization":
            c = l.weights["gamma"].size
            l.weights["gamma"] = rng.uniform(0.7, 1.3, c).astype(np.float32)
            l.weights["beta"] = rng.normal(0, 0.3, c).astype(np.float32)
            l.weights["moving_mean"] = rng.normal(0, 0.2, c).astype(np.float32)
            l.weights["moving_variance"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    inference = builder.get_model_for_inference(model_trained=model, max_number_of_boxes_per_class=4, max_number_of_boxes_per_sample=10,
                                                boxes_iou_threshold=0.5, labels_probability_threshold=0.3, suppress_background_boxes=False,
                                                use_segmentation_suppression=True)
    x = rng.integers(0, 256, (1, 480, 640, 3)).astype(np.float32)
    seg, det = inference(x, training=False)                 # NB03#cell31 call syntax
    assert seg.shape == (1, 480, 640, 4) and det.shape == (1, 10, 6)
    assert np.isfinite(seg).all() and np.isfinite(det).all()
    assert np.abs(seg.sum(-1) - 1).max() < 1e-5

    ref = NpModel(inference, dtype=np.float32)
    ref.set_weights_from(lambda l: l.get_weights())
    seg_ref, det_ref = ref.forward(x, training=False)
    assert seg_ref.shape == seg.shape and det_ref.shape == det.shape
    assert np.abs(seg - seg_ref).max() < 1e-3
    eng = E.engine_for(inference, 1, False)
    probs = eng.vals[id(inference.get_layer('output-labels').outputs[0])].store.buf.download().reshape(1, 9600, 4)
    offs = eng.vals[id(inference.get_layer('output-boxes').outputs[0])].store.buf.download().reshape(1, 9600, 4)
    assert np.abs(probs.sum(-1) - 1).max() < 1e-5
    assert np.abs(probs - ref.value('output-labels')).max() < 1e-3
    assert np.abs(offs - ref.value('output-boxes')).max() < 1e-3 * max(1.0, np.abs(ref.value('output-boxes')).max())
    # inference tail: exact given the same head tensors (index kernels), close to the all-oracle result
    dec = inference.get_layer('decode-output-boxes')
    cent = np.stack([dec.center_x_boxes_default, dec.center_y_boxes_default, dec.width_boxes_default, dec.height_boxes_default], axis=1)
    corners = O.decode_to_corners_pred(offs, cent, bench.STDS)
    want, valid = O.combined_nms(corners, O.seg_suppress(seg, probs), 4, 10, 0.5, 0.3)
    assert np.array_equal(det[..., 0], want[..., 0]) and np.abs(det - want).max() < 1e-3
    assert valid[0] > 0, "thresholds chosen so that the random-weight model emits detections"
