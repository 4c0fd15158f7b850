"""Custom layers of the inference tail -- same constructor kwargs as the reference (layers.py:6-17, :97-105, :181,
:216).  As graph nodes they are lowered by `_engine.py` to the HIP kernels ssdseg_decode_boxes /
ssdseg_combined_nms / ssdseg_seg_suppress; `Split` is a zero-copy channel view.
"""
from typing import List, Union

import numpy as np

from . import _graph as K


class DecodeBoxesCentroidsOffsets(K.Layer):
    """Predicted centroid offsets -> corners (ymin, xmin, ymax, xmax) (reference layers.py:45-81)."""
    type_name = "DecodeBoxesCentroidsOffsets"

    def __init__(self, center_x_boxes_default, center_y_boxes_default, width_boxes_default, height_boxes_default,
                 standard_deviation_center_x_offsets: float, standard_deviation_center_y_offsets: float,
                 standard_deviation_width_offsets: float, standard_deviation_height_offsets: float, **kwargs):
        super().__init__(**kwargs)
        self.center_x_boxes_default = np.asarray(center_x_boxes_default, np.float32)
        self.center_y_boxes_default = np.asarray(center_y_boxes_default, np.float32)
        self.width_boxes_default = np.asarray(width_boxes_default, np.float32)
        self.height_boxes_default = np.asarray(height_boxes_default, np.float32)
        self.standard_deviation_center_x_offsets = float(standard_deviation_center_x_offsets)
        self.standard_deviation_center_y_offsets = float(standard_deviation_center_y_offsets)
        self.standard_deviation_width_offsets = float(standard_deviation_width_offsets)
        self.standard_deviation_height_offsets = float(standard_deviation_height_offsets)

    def get_config(self):
        return {k: getattr(self, k) for k in (
            'center_x_boxes_default', 'center_y_boxes_default', 'width_boxes_default', 'height_boxes_default',
            'standard_deviation_center_x_offsets', 'standard_deviation_center_y_offsets',
            'standard_deviation_width_offsets', 'standard_deviation_height_offsets')}


class NonMaximumSuppression(K.Layer):
    """Combined per-class NMS + repack to (label, prob, xmin, ymin, xmax, ymax) (reference layers.py:127-168)."""
    type_name = "NonMaximumSuppression"

    def __init__(self, max_number_of_boxes_per_class: int, max_number_of_boxes_per_sample: int, boxes_iou_threshold: float,
                 labels_probability_threshold: float, suppress_background_boxes: bool, **kwargs):
        super().__init__(**kwargs)
        self.max_number_of_boxes_per_class = max_number_of_boxes_per_class
        self.max_number_of_boxes_per_sample = max_number_of_boxes_per_sample
        self.boxes_iou_threshold = boxes_iou_threshold
        self.labels_probability_threshold = labels_probability_threshold
        self.suppress_background_boxes = suppress_background_boxes

    def build(self, input_shapes):
        return (None, self.max_number_of_boxes_per_sample, 6)

    def get_config(self):
        return {k: getattr(self, k) for k in (
            'max_number_of_boxes_per_class', 'max_number_of_boxes_per_sample', 'boxes_iou_threshold',
            'labels_probability_threshold', 'suppress_background_boxes')}


class SegmentationSuppression(K.Layer):
    """Zero the probabilities of classes the segmentation head did not predict anywhere in the batch
    (reference layers.py:189-212; depth hard-coded to 4 there, quirk Q6)."""
    type_name = "SegmentationSuppression"

    def build(self, input_shapes):
        return input_shapes[1]


class Split(K.Layer):
    """tf.split wrapper (reference layers.py:214-244); only even channel splits are used (models.py:573)."""
    type_name = "Split"

    def __init__(self, num_or_size_splits: Union[int, List[int]], axis: int, num: int = None, **kwargs):
        super().__init__(**kwargs)
        self.num_or_size_splits = num_or_size_splits
        self.axis = axis
        self.num = num

    def build(self, input_shapes):
        s = list(input_shapes[0])
        ax = self.axis % len(s)
        sizes = [s[ax] // self.num_or_size_splits] * self.num_or_size_splits if isinstance(self.num_or_size_splits, int) else list(self.num_or_size_splits)
        assert sum(sizes) == s[ax]
        return [tuple(s[:ax] + [z] + s[ax + 1:]) for z in sizes]

    def get_config(self):
        # the reference's get_config reads a misspelt attribute and would raise (quirk Q9); fixed here
        return {'num_or_size_splits': self.num_or_size_splits, 'axis': self.axis, 'num': self.num}
