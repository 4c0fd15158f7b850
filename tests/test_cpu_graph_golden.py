"""The builders reproduce the reference graph: every row of the reference's own `model.summary()` output
(NB03#cell12, parsed into tests/golden/nb03_model_summary.json) -- name, type, shape, #params, inbound layers, order."""
import json

import numpy as np
import pytest


def _builder(kind="mobilenet", **kw):
    import ssdseglib
    d = np.zeros(9600, np.float32)
    common = dict(number_of_boxes_per_point=[6, 6, 6, 6], number_of_classes=4, center_x_boxes_default=d, center_y_boxes_default=d,
                  width_boxes_default=d, height_boxes_default=d, standard_deviations_centroids_offsets=(0.1, 0.1, 0.2, 0.2))
    if kind == "mobilenet":
        return ssdseglib.models.MobileNetV2SsdSegBuilder(input_image_shape=(480, 640, 3), **common)
    return ssdseglib.models.ShuffleNetV2SsdSegBuilder(input_image_shape=(480, 640, 3), **kw, **common)


def test_mobilenetv2_summary_matches_reference_output(golden_dir):
    gold = json.load(open(f"{golden_dir}/nb03_model_summary.json"))
    model = _builder().get_model_for_training('deeplabv3plus', 'ssdlite', segmentation_dilation_rates=(3, 6, 12))
    assert len(model.layers) == len(gold["layers"]) == 224
    for layer, g in zip(model.layers, gold["layers"]):
        shape = layer.output_shape
        shape = shape[0] if isinstance(shape, list) else shape
        assert layer.name == g["name"]
        assert layer.type_name == g["type"], layer.name
        assert list(shape) == g["output_shape"], layer.name
        assert layer.count_params() == g["params"], layer.name
        assert [t.layer.name for t in layer.inbound] == g["inbound"], layer.name
    assert model.count_params() == gold["totals"]["Total params"] == 4047408
    assert sum(l.count_trainable() for l in model.layers) == gold["totals"]["Trainable params"] == 4009920
    assert model.output_names == ['output-mask', 'output-labels', 'output-boxes']
    lines = []
    model.summary(print_fn=lines.append)
    assert any("Total params: 4047408" in l for l in lines)


@pytest.mark.parametrize("size,params", [("1x", 2790118), ("1.5x", 4572328)])     # SURVEY.md App. D counts
def test_shufflenetv2_variants(size, params):
    model = _builder("shufflenet", model_size=size, use_additional_depthwise_convolution=True,
                     use_residual_connections=True).get_model_for_training('deeplabv3plus', 'ssdlite', (3, 6, 12))
    assert model.count_params() == params
    assert [tuple(t.shape) for t in model.outputs] == [(None, 480, 640, 4), (None, 9600, 4), (None, 9600, 4)]
    # quirk Q1: the heads use ReLU(max_value=0.0) because the blocks are called with their default
    assert model.get_layer('labels1-relu0').max_value == 0.0 and model.get_layer('mask-decoder-conv-relu0').max_value == 0.0


def test_shufflenetv2_bad_size_raises():
    with pytest.raises(ValueError):
        _builder("shufflenet", model_size="3x", use_additional_depthwise_convolution=False, use_residual_connections=False)


def test_inference_graph_and_weight_api():
    b = _builder()
    model = b.get_model_for_training('deeplabv3plus', 'ssdlite', (3, 6, 12))
    inf = b.get_model_for_inference(model, 4, 10, 0.025, 0.725, False, True)
    assert [tuple(t.shape) for t in inf.outputs] == [(None, 480, 640, 4), (None, 10, 6)]
    assert inf.get_layer('segmentation-suppression').type_name == 'SegmentationSuppression'
    conv = model.get_layer('backbone-block1-expand-conv')
    (k,) = conv.get_weights()
    assert k.shape == (1, 1, 16, 96)
    conv.set_weights([k * 2])
    assert np.array_equal(conv.get_weights()[0], k * 2)
    with pytest.raises(ValueError):
        conv.set_weights([k[..., :3]])
    with pytest.raises(ValueError):
        model.get_layer('no-such-layer')
    sep = model.get_layer('mask-decoder-sepconv')
    assert [w.shape for w in sep.get_weights()] == [(3, 3, 256, 1), (1, 1, 256, 256)]        # Keras order: depthwise, pointwise
    bn = model.get_layer('backbone-block0-expand-batchnorm')
    assert list(bn.weights) == ['gamma', 'beta', 'moving_mean', 'moving_variance']
