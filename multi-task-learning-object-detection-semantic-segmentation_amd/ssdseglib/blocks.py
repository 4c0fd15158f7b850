"""DeepLabV3+ encoder/decoder and SSDLite head blocks -- graph-building functions with the reference's
signatures (reference blocks.py:4, :76, :134).  They only describe the graph (layer names, shapes, Keras weight
layouts); `_engine.py` lowers each conv+BN+ReLU chain to fused HIP launches.

Quirk Q1 is kept: `relu_max_value` defaults to 0.0, and Keras' ReLU(max_value=0.0) outputs zeros -- the
ShuffleNetV2 builder (like the reference, models.py:685-758) calls these blocks without overriding it.
"""
from typing import Tuple

from . import _graph as K


def _conv_bn_relu(layer, conv, prefix: str, relu_max_value: float):
    """conv -> BatchNormalization -> ReLU(max_value) with the reference's naming scheme."""
    layer = conv(layer)
    layer = K.BatchNormalization(name=f'{prefix}batchnorm')(layer)
    return K.ReLU(max_value=relu_max_value, name=f'{prefix}relu{int(relu_max_value)}')(layer)


def deeplabv3plus_encoder(layer, filters: int = 256, dilation_rates: Tuple[int, int, int] = (6, 12, 18), relu_max_value: float = 0.0):
    """ASPP + image-pooling branch + 1x1 fuse (reference blocks.py:24-74)."""
    aspp = 'mask-encoder-aspp-'
    branches = [_conv_bn_relu(layer, K.Conv2D(filters=filters, kernel_size=1, padding='same', use_bias=False, name=f'{aspp}pointwise-conv'),
                              f'{aspp}pointwise-', relu_max_value)]
    for i, rate in enumerate(dilation_rates, start=1):
        sep = K.SeparableConv2D(filters=filters, kernel_size=3, padding='same', dilation_rate=rate, depth_multiplier=1,
                                use_bias=False, name=f'{aspp}atrous{i}-sepconv')
        branches.append(_conv_bn_relu(layer, sep, f'{aspp}atrous{i}-', relu_max_value))

    pool = 'mask-encoder-pooling-'
    size = tuple(layer.shape[1:3])
    pooled = K.GlobalAveragePooling2D(data_format='channels_last', keepdims=True, name=f'{pool}globalavgpool')(layer)
    pooled = K.Conv2D(filters=filters, kernel_size=1, padding='same', use_bias=False, name=f'{pool}conv')(pooled)
    pooled = K.BatchNormalization(name=f'{pool}batchnorm')(pooled)
    pooled = K.ReLU(max_value=relu_max_value, name=f'{pool}relu{int(relu_max_value)}')(pooled)
    pooled = K.UpSampling2D(size=size, interpolation='bilinear', name=f'{pool}upsampling')(pooled)
    branches.append(pooled)

    enc = 'mask-encoder-'
    fused = K.Concatenate(axis=-1, name=f'{enc}concat')(branches)
    fused = K.Conv2D(filters=filters, kernel_size=1, padding='same', use_bias=False, name=f'{enc}output-conv')(fused)
    fused = K.BatchNormalization(name=f'{enc}output-batchnorm')(fused)
    return K.ReLU(max_value=relu_max_value, name=f'{enc}output-relu{int(relu_max_value)}')(fused)


def deeplabv3plus_decoder(layer_encoder, layer_backbone, filters_backbone: int, filters_decoder: int,
                          output_height_width: Tuple[int, int], output_channels: int, relu_max_value: float = 0.0):
    """Upsample encoder, reduce + concat the backbone tap, refine, classify, upsample, softmax (reference blocks.py:100-132)."""
    dec = 'mask-decoder-'
    up = (int(layer_backbone.shape[1] / layer_encoder.shape[1]), int(layer_backbone.shape[2] / layer_encoder.shape[2]))
    layer_encoder = K.UpSampling2D(size=up, interpolation='bilinear', name=f'{dec}upsampling-encoder-output')(layer_encoder)
    if filters_backbone is not None:
        layer_backbone = K.Conv2D(filters=filters_backbone, kernel_size=1, padding='same', use_bias=False, name=f'{dec}backbone-conv')(layer_backbone)
        layer_backbone = K.BatchNormalization(name=f'{dec}backbone-batchnorm')(layer_backbone)
        layer_backbone = K.ReLU(max_value=relu_max_value, name=f'{dec}backbone-relu{int(relu_max_value)}')(layer_backbone)
    layer = K.Concatenate(axis=-1, name=f'{dec}concat')([layer_encoder, layer_backbone])

    layer = K.Conv2D(filters=filters_decoder, kernel_size=3, padding='same', use_bias=False, name=f'{dec}conv')(layer)
    layer = K.BatchNormalization(name=f'{dec}conv-batchnorm')(layer)
    layer = K.ReLU(max_value=relu_max_value, name=f'{dec}conv-relu{int(relu_max_value)}')(layer)

    layer = K.SeparableConv2D(filters=filters_decoder, kernel_size=3, padding='same', depth_multiplier=1, use_bias=False, name=f'{dec}sepconv')(layer)
    layer = K.BatchNormalization(name=f'{dec}sepconv-batchnorm')(layer)
    layer = K.ReLU(max_value=relu_max_value, name=f'{dec}sepconv-relu{int(relu_max_value)}')(layer)

    layer = K.Conv2D(filters=output_channels, kernel_size=3, padding='same', use_bias=False, name=f'{dec}output-conv')(layer)
    up_out = (int(output_height_width[0] / layer.shape[1]), int(output_height_width[1] / layer.shape[2]))
    layer = K.UpSampling2D(size=up_out, interpolation='bilinear', name=f'{dec}output-upsampling')(layer)
    return K.Softmax(name='output-mask')(layer)


def ssdlite(layer, filters: int, output_channels: int, name_prefix: str, relu_max_value: float = 0.0):
    """SeparableConv2D 3x3 -> BN -> ReLU -> Reshape(-1, output_channels) (reference blocks.py:152-157)."""
    layer = K.SeparableConv2D(filters=filters, kernel_size=3, padding='same', depth_multiplier=1, use_bias=False, name=f'{name_prefix}sepconv')(layer)
    layer = K.BatchNormalization(name=f'{name_prefix}batchnorm')(layer)
    layer = K.ReLU(max_value=relu_max_value, name=f'{name_prefix}relu{int(relu_max_value)}')(layer)
    return K.Reshape(target_shape=(-1, output_channels), name=f'{name_prefix}reshape')(layer)
