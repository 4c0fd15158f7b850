"""Model builders with the reference's API (`MobileNetV2SsdSegBuilder` reference models.py:6-423,
`ShuffleNetV2SsdSegBuilder` :425-870): a backbone, a DeepLabV3+ segmentation head and an SSDLite detection head,
three training outputs `output-mask`, `output-labels`, `output-boxes`, every layer named as in NB03#cell12.

The builders only describe the graph (`_graph.py`); running it (`model(x)`, `fit`, `predict`) goes through
`_engine.py`, which lowers conv+BN+ReLU chains to the fused gfx950 kernels behind include/ssdseg.h.
"""
from typing import List, Literal, Tuple, Union

from numpy import ndarray

from . import _graph as K
from . import blocks, layers

# (expansion, output channels, repeats, first stride) of the MobileNetV2 bottleneck sequences (reference models.py:205-210)
_MOBILENETV2_SEQUENCES = ((6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))

# stage -> channels per ShuffleNetV2 size (reference models.py:459-466)
_SHUFFLENETV2_CHANNELS = {'0.5x': {2: 48, 3: 96, 4: 192}, '1x': {2: 116, 3: 232, 4: 464},
                          '1.5x': {2: 176, 3: 352, 4: 704}, '2x': {2: 244, 3: 488, 4: 976}}


class _SsdSegBuilderBase:
    """State and head construction shared by both backbones."""

    # names of the four SSD feature maps / encoder input / decoder tap are set by the subclasses
    relu_max_value_heads = None  # None -> call the blocks with their default (quirk Q1)

    def _init_common(self, input_image_shape, number_of_boxes_per_point, number_of_classes, center_x_boxes_default,
                     center_y_boxes_default, width_boxes_default, height_boxes_default, standard_deviations_centroids_offsets):
        self.input_image_shape = input_image_shape
        self.number_of_boxes_per_point = (number_of_boxes_per_point,) * 4 if isinstance(number_of_boxes_per_point, int) else number_of_boxes_per_point
        self.number_of_classes = number_of_classes
        self._center_x_boxes_default = center_x_boxes_default
        self._center_y_boxes_default = center_y_boxes_default
        self._width_boxes_default = width_boxes_default
        self._height_boxes_default = height_boxes_default
        (self._standard_deviation_center_x_offsets, self._standard_deviation_center_y_offsets,
         self._standard_deviation_width_offsets, self._standard_deviation_height_offsets) = standard_deviations_centroids_offsets
        self._layers = {}

    def _block_kwargs(self):
        return {} if self.relu_max_value_heads is None else {'relu_max_value': self.relu_max_value_heads}

    def _register_layers(self, model: K.Model):
        self._layers = {layer.name: layer.output for layer in model.layers}

    def _ssd_branches(self, feature_maps, prefix: str, values_per_box: int):
        """one ssdlite block per feature map -> list of (None, boxes, values_per_box) tensors"""
        return [blocks.ssdlite(layer=fm, filters=self.number_of_boxes_per_point[i] * values_per_box, output_channels=values_per_box,
                               name_prefix=f'{prefix}{i + 1}-', **self._block_kwargs())
                for i, fm in enumerate(feature_maps)]

    def _ssd_outputs(self, feature_maps):
        # classification: the reference hard-codes 4 here and uses number_of_classes for the boxes (quirk Q2,
        # reference models.py:250-253 vs :265-268); kept, it only works because classes == coordinates == 4
        labels = K.Concatenate(axis=1, name='labels-concat')(self._ssd_branches(feature_maps, 'labels', 4))
        output_labels = K.Softmax(name='output-labels')(labels)
        output_boxes = K.Concatenate(axis=1, name='output-boxes')(self._ssd_branches(feature_maps, 'boxes', self.number_of_classes))
        return output_labels, output_boxes

    def _assemble_training_model(self, layer_input, segmentation_architecture, object_detection_architecture, segmentation_dilation_rates):
        if segmentation_architecture == 'deeplabv3plus':
            output_mask = self._semantic_segmentation_head_deeplabv3plus(dilation_rates=segmentation_dilation_rates)
        if object_detection_architecture == 'ssdlite':
            output_labels, output_boxes = self._object_detection_head_ssdlite()
        model = K.Model(inputs=layer_input, outputs=[output_mask, output_labels, output_boxes], name='model_1')
        self._register_layers(model)
        return model

    def get_model_for_inference(self, model_trained: K.Model, max_number_of_boxes_per_class: int, max_number_of_boxes_per_sample: int,
                                boxes_iou_threshold: float, labels_probability_threshold: float, suppress_background_boxes: bool,
                                use_segmentation_suppression: bool) -> K.Model:
        """Trained graph + decode + (segmentation suppression) + combined NMS; outputs [mask, detections]
        (reference models.py:345-423 / :792-870).  The inference model shares the trained model's layers, so the
        reference's layer-by-layer weight copy (:420-421) is the identity here."""
        layer_input = model_trained.get_layer('backbone-input').output
        output_mask = model_trained.get_layer('output-mask').output
        boxes = model_trained.get_layer('output-boxes').output
        labels = model_trained.get_layer('output-labels').output

        decode = layers.DecodeBoxesCentroidsOffsets(
            center_x_boxes_default=self._center_x_boxes_default, center_y_boxes_default=self._center_y_boxes_default,
            width_boxes_default=self._width_boxes_default, height_boxes_default=self._height_boxes_default,
            standard_deviation_center_x_offsets=self._standard_deviation_center_x_offsets,
            standard_deviation_center_y_offsets=self._standard_deviation_center_y_offsets,
            standard_deviation_width_offsets=self._standard_deviation_width_offsets,
            standard_deviation_height_offsets=self._standard_deviation_height_offsets, name='decode-output-boxes')
        decode.trainable = False
        nms = layers.NonMaximumSuppression(
            max_number_of_boxes_per_class=max_number_of_boxes_per_class, max_number_of_boxes_per_sample=max_number_of_boxes_per_sample,
            boxes_iou_threshold=boxes_iou_threshold, labels_probability_threshold=labels_probability_threshold,
            suppress_background_boxes=suppress_background_boxes, name='output-object-detection')
        nms.trainable = False
        if use_segmentation_suppression:
            suppression = layers.SegmentationSuppression(name='segmentation-suppression')
            suppression.trainable = False
            labels = suppression(segmentation_mask=output_mask, labels_probabilities=labels)
        detections = nms(boxes_corners_coordinates=decode(boxes), labels_probabilities=labels)
        model = K.Model(inputs=layer_input, outputs=[output_mask, detections], name='model_inference')
        model._trained = model_trained
        return model


class MobileNetV2SsdSegBuilder(_SsdSegBuilderBase):
    relu_max_value_heads = 6.0

    def __init__(self, input_image_shape: Tuple[int, int, int], number_of_boxes_per_point: Union[int, List[int]], number_of_classes: int,
                 center_x_boxes_default: ndarray, center_y_boxes_default: ndarray, width_boxes_default: ndarray,
                 height_boxes_default: ndarray, standard_deviations_centroids_offsets: tuple) -> None:
        self._init_common(input_image_shape, number_of_boxes_per_point, number_of_classes, center_x_boxes_default, center_y_boxes_default,
                          width_boxes_default, height_boxes_default, standard_deviations_centroids_offsets)
        self._counter_blocks = 0

    # ---- the three MobileNetV2 building blocks (reference models.py:47-113)
    def _mobilenetv2_block_expand(self, layer, channels: int, kernel_size=1, strides=1):
        p = f'backbone-block{self._counter_blocks}-expand-'
        layer = K.Conv2D(filters=channels, kernel_size=kernel_size, strides=strides, padding='same', use_bias=False, name=f'{p}conv')(layer)
        layer = K.BatchNormalization(name=f'{p}batchnorm')(layer)
        return K.ReLU(max_value=6.0, name=f'{p}relu6')(layer)

    def _mobilenetv2_block_depthwise(self, layer, strides):
        p = f'backbone-block{self._counter_blocks}-depthwise-'
        layer = K.DepthwiseConv2D(depth_multiplier=1, kernel_size=3, strides=strides, padding='same', use_bias=False, name=f'{p}conv')(layer)
        layer = K.BatchNormalization(name=f'{p}batchnorm')(layer)
        return K.ReLU(max_value=6.0, name=f'{p}relu6')(layer)

    def _mobilenetv2_block_project(self, layer, channels: int):
        p = f'backbone-block{self._counter_blocks}-project-'
        layer = K.Conv2D(filters=channels, kernel_size=1, padding='same', use_bias=False, name=f'{p}conv')(layer)
        return K.BatchNormalization(name=f'{p}batchnorm')(layer)

    def _mobilenetv2_block_sequence(self, layer, expansion_factor: int, channels_output: int, n_repeat: int, strides):
        """n_repeat inverted residuals; stride only on the first, residual Add from the second on (reference models.py:135-167)."""
        carried = layer
        for n in range(n_repeat):
            self._counter_blocks += 1
            x = self._mobilenetv2_block_expand(layer=carried, channels=carried.shape[-1] * expansion_factor)
            x = self._mobilenetv2_block_depthwise(layer=x, strides=strides if n == 0 else 1)
            x = self._mobilenetv2_block_project(layer=x, channels=channels_output)
            carried = x if n == 0 else K.Add(name=f'backbone-block{self._counter_blocks}-add')([carried, x])
        return carried

    def _mobilenetv2_backbone(self):
        layer_input = K.Input(shape=self.input_image_shape, dtype='float32', name='backbone-input')
        layer = K.Rescaling(scale=1. / 127.5, offset=-1, name='backbone-input-rescaling')(layer_input)
        # block 0: 3x3 stride-2 "expand" to 32, depthwise, project to 16 (reference models.py:196-202)
        layer = self._mobilenetv2_block_expand(layer=layer, channels=32, kernel_size=3, strides=2)
        layer = self._mobilenetv2_block_depthwise(layer=layer, strides=1)
        layer = self._mobilenetv2_block_project(layer=layer, channels=16)
        for expansion, channels, repeats, stride in _MOBILENETV2_SEQUENCES:
            layer = self._mobilenetv2_block_sequence(layer=layer, expansion_factor=expansion, channels_output=channels, n_repeat=repeats, strides=stride)
        self._register_layers(K.Model(inputs=layer_input, outputs=layer))
        return layer_input

    def _extra_feature_map(self, layer, filters: int):
        self._counter_blocks += 1
        p = f'backbone-block{self._counter_blocks}-'
        layer = K.SeparableConv2D(filters=filters, strides=2, kernel_size=3, padding='same', depth_multiplier=1, use_bias=False, name=f'{p}sepconv')(layer)
        layer = K.BatchNormalization(name=f'{p}batchnorm')(layer)
        return K.ReLU(max_value=6.0, name=f'{p}relu6')(layer)

    def _object_detection_head_ssdlite(self):
        fm1 = self._layers['backbone-block13-expand-relu6']
        fm2 = self._layers['backbone-block16-project-batchnorm']
        fm3 = self._extra_feature_map(fm2, 320)      # reference models.py:234-238
        fm4 = self._extra_feature_map(fm3, 360)      # reference models.py:240-244
        return self._ssd_outputs([fm1, fm2, fm3, fm4])

    def _semantic_segmentation_head_deeplabv3plus(self, dilation_rates: Tuple[int, int, int] = (6, 12, 18)):
        encoder = blocks.deeplabv3plus_encoder(layer=self._layers['backbone-block13-expand-relu6'], filters=256,
                                               dilation_rates=dilation_rates, relu_max_value=6.0)
        return blocks.deeplabv3plus_decoder(layer_encoder=encoder, layer_backbone=self._layers['backbone-block3-expand-relu6'],
                                            filters_backbone=48, filters_decoder=256, output_height_width=self.input_image_shape[0:2],
                                            output_channels=self.number_of_classes, relu_max_value=6.0)

    def get_model_for_training(self, segmentation_architecture: Literal['deeplabv3plus'], object_detection_architecture: Literal['ssdlite'],
                               segmentation_dilation_rates: Tuple[int, int, int] = (6, 12, 18)) -> K.Model:
        self._counter_blocks = 0
        layer_input = self._mobilenetv2_backbone()
        return self._assemble_training_model(layer_input, segmentation_architecture, object_detection_architecture, segmentation_dilation_rates)


class ShuffleNetV2SsdSegBuilder(_SsdSegBuilderBase):
    relu_max_value_heads = None  # quirk Q1: blocks are called with their default relu_max_value=0.0

    def __init__(self, input_image_shape: Tuple[int, int, int], model_size: Literal['0.5x', '1x', '1.5x', '2x'],
                 use_additional_depthwise_convolution: bool, use_residual_connections: bool,
                 number_of_boxes_per_point: Union[int, List[int]], number_of_classes: int, center_x_boxes_default: ndarray,
                 center_y_boxes_default: ndarray, width_boxes_default: ndarray, height_boxes_default: ndarray,
                 standard_deviations_centroids_offsets: tuple) -> None:
        if model_size not in _SHUFFLENETV2_CHANNELS:
            raise ValueError('invalid "model_size" value! available values are "0.5x", "1x", "1.5x", "2x"')
        self.output_channels_stages = dict(_SHUFFLENETV2_CHANNELS[model_size])
        self.use_additional_depthwise_convolution = use_additional_depthwise_convolution
        self.use_residual_connections = use_residual_connections
        self._init_common(input_image_shape, number_of_boxes_per_point, number_of_classes, center_x_boxes_default, center_y_boxes_default,
                          width_boxes_default, height_boxes_default, standard_deviations_centroids_offsets)

    def _shufflenetv2_block_channels_shuffle(self, layer, name_prefix: str, groups: int = 2):
        """Reshape -> Permute -> Reshape (reference models.py:494-505)."""
        _, height, width, channels = layer.get_shape().as_list()
        layer = K.Reshape(target_shape=(height, width, groups, channels // groups), name=f'{name_prefix}reshape-pre-channels-shuffle')(layer)
        layer = K.Permute(dims=(1, 2, 4, 3), name=f'{name_prefix}channels-shuffle')(layer)
        return K.Reshape(target_shape=(height, width, channels), name=f'{name_prefix}reshape-post-channels-shuffle')(layer)

    @staticmethod
    def _dw_bn(layer, name_conv, name_bn, strides=1):
        layer = K.DepthwiseConv2D(kernel_size=3, strides=strides, padding='same', depth_multiplier=1, use_bias=False, name=name_conv)(layer)
        return K.BatchNormalization(name=name_bn)(layer)

    @staticmethod
    def _pw_bn(layer, filters, name_conv, name_bn):
        layer = K.Conv2D(filters=filters, kernel_size=1, padding='same', use_bias=False, name=name_conv)(layer)
        return K.BatchNormalization(name=name_bn)(layer)

    def _shufflenetv2_block_downsampling_unit(self, layer, output_channels: int, name_prefix: str):
        """two stride-2 branches, concat, shuffle (reference models.py:520-555)."""
        filters = output_channels // 2
        left, right = f'{name_prefix}branch-left-', f'{name_prefix}branch-right-'
        bl = self._dw_bn(layer, f'{left}depthconv1', f'{left}batchnorm1', strides=2)
        bl = self._pw_bn(bl, filters, f'{left}conv2', f'{left}batchnorm2')
        bl = K.ReLU(name=f'{left}relu2')(bl)

        br = layer
        if self.use_additional_depthwise_convolution:
            br = self._dw_bn(br, f'{right}depthconv0', f'{right}batchnorm0')
        br = self._pw_bn(br, filters, f'{right}conv1', f'{right}batchnorm1')
        br = K.ReLU(name=f'{right}relu1')(br)
        br = self._dw_bn(br, f'{right}depthconv2', f'{right}batchnorm2', strides=2)
        br = self._pw_bn(br, filters, f'{right}conv3', f'{right}batchnorm3')
        br = K.ReLU(name=f'{right}relu3')(br)

        merged = K.Concatenate(axis=-1, name=f'{name_prefix}-concat')([bl, br])
        return self._shufflenetv2_block_channels_shuffle(merged, name_prefix=f'{name_prefix}')

    def _shufflenetv2_block_basic_unit(self, layer, output_channels: int, name_prefix: str):
        """split, conv branch (+dw, +residual), concat, shuffle (reference models.py:570-603)."""
        filters = output_channels // 2
        conv = f'{name_prefix}branch-conv-'
        identity, half = layers.Split(num_or_size_splits=2, axis=-1, name=f'{name_prefix}channels-split')(layer)
        b = half
        if self.use_additional_depthwise_convolution:
            b = self._dw_bn(b, f'{conv}depthconv0', f'{conv}batchnorm0')
        b = self._pw_bn(b, filters, f'{conv}conv1', f'{conv}batchnorm1')
        b = K.ReLU(name=f'{conv}relu1')(b)
        b = self._dw_bn(b, f'{conv}depthconv2', f'{conv}batchnorm2')
        b = self._pw_bn(b, filters, f'{conv}conv3', f'{conv}batchnorm3')
        if self.use_residual_connections:
            b = K.Add(name=f'{conv}add')([b, half])
        b = K.ReLU(name=f'{conv}relu3')(b)
        merged = K.Concatenate(axis=-1, name=f'{name_prefix}concat')([identity, b])
        return self._shufflenetv2_block_channels_shuffle(merged, name_prefix=f'{name_prefix}')

    def _shufflenetv2_backbone(self):
        layer_input = K.Input(shape=self.input_image_shape, dtype='float32', name='backbone-input')
        layer = K.Rescaling(scale=1. / 127.5, offset=-1, name='backbone-input-rescaling')(layer_input)
        layer = K.Conv2D(filters=24, kernel_size=3, strides=2, padding='same', use_bias=True, name='backbone-stage1-conv')(layer)
        layer = K.MaxPooling2D(pool_size=3, strides=2, padding='same', name='backbone-stage1-maxpool')(layer)
        for stage, units in ((2, 3), (3, 7), (4, 3)):     # reference models.py:632-647
            channels = self.output_channels_stages[stage]
            layer = self._shufflenetv2_block_downsampling_unit(layer, output_channels=channels, name_prefix=f'backbone-stage{stage}-downblock-')
            for unit in range(units):
                layer = self._shufflenetv2_block_basic_unit(layer, output_channels=channels, name_prefix=f'backbone-stage{stage}-block{unit + 1}-')
        self._register_layers(K.Model(inputs=layer_input, outputs=layer))
        return layer_input

    def _extra_feature_map(self, layer, name_prefix: str):
        layer = K.SeparableConv2D(filters=self.output_channels_stages[4], strides=2, kernel_size=3, padding='same', depth_multiplier=1,
                                  use_bias=False, name=f'{name_prefix}sepconv')(layer)
        layer = K.BatchNormalization(name=f'{name_prefix}batchnorm')(layer)
        return K.ReLU(name=f'{name_prefix}relu')(layer)

    def _object_detection_head_ssdlite(self):
        fm1 = self._layers['backbone-stage3-block7-reshape-post-channels-shuffle']
        fm2 = self._layers['backbone-stage4-block3-reshape-post-channels-shuffle']
        fm3 = self._extra_feature_map(fm2, 'backbone-stage5-block1-')
        fm4 = self._extra_feature_map(fm3, 'backbone-stage5-block2-')
        return self._ssd_outputs([fm1, fm2, fm3, fm4])

    def _semantic_segmentation_head_deeplabv3plus(self, dilation_rates: Tuple[int, int, int] = (6, 12, 18)):
        encoder = blocks.deeplabv3plus_encoder(layer=self._layers['backbone-stage3-block7-reshape-post-channels-shuffle'],
                                               filters=256, dilation_rates=dilation_rates)
        return blocks.deeplabv3plus_decoder(layer_encoder=encoder,
                                            layer_backbone=self._layers['backbone-stage2-block3-reshape-post-channels-shuffle'],
                                            filters_backbone=48, filters_decoder=256, output_height_width=self.input_image_shape[0:2],
                                            output_channels=self.number_of_classes)

    def get_model_for_training(self, segmentation_architecture: Literal['deeplabv3plus'], object_detection_architecture: Literal['ssdlite'],
                               segmentation_dilation_rates: Tuple[int, int, int] = (6, 12, 18)) -> K.Model:
        layer_input = self._shufflenetv2_backbone()
        return self._assemble_training_model(layer_input, segmentation_architecture, object_detection_architecture, segmentation_dilation_rates)
