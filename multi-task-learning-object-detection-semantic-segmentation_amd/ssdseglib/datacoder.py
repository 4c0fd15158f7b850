"""Ground-truth encoder / decoder with the reference's API (reference datacoder.py:5-432).

The matching + offset encoding (the hot part: IoU 9600 x G, three-step matching, scatter) runs on the GPU through
ssdseg_encode_targets, for one sample (`read_and_encode`, like the reference's tf.data map) or a whole batch
(`encode_batch`, what the training step uses).  File reading is host glue: CSV via NumPy, PNG via Pillow.
"""
import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

_CORNERS = ("xmin_boxes_default", "ymin_boxes_default", "xmax_boxes_default", "ymax_boxes_default")
_CENTROIDS = ("center_x_boxes_default", "center_y_boxes_default", "width_boxes_default", "height_boxes_default")


class DataEncoderDecoder:
    def __init__(self, num_classes: int, image_shape: Tuple[int, int],
                 xmin_boxes_default=None, ymin_boxes_default=None, xmax_boxes_default=None, ymax_boxes_default=None,
                 center_x_boxes_default=None, center_y_boxes_default=None, width_boxes_default=None, height_boxes_default=None,
                 iou_threshold: float = 0.5, standard_deviations_centroids_offsets: Tuple[float, ...] = (0.1, 0.1, 0.2, 0.2),
                 augmentation_horizontal_flip: bool = False) -> None:
        self.num_classes = num_classes
        self.image_height, self.image_width = image_shape
        self.iou_threshold = iou_threshold
        (self.standard_deviation_center_x_offsets, self.standard_deviation_center_y_offsets,
         self.standard_deviation_width_offsets, self.standard_deviation_height_offsets) = standard_deviations_centroids_offsets
        corners = (xmin_boxes_default, ymin_boxes_default, xmax_boxes_default, ymax_boxes_default)
        centroids = (center_x_boxes_default, center_y_boxes_default, width_boxes_default, height_boxes_default)
        f32 = lambda a: np.asarray(a, dtype=np.float32)
        if all(v is None for v in centroids):
            if any(v is None for v in corners):
                raise ValueError('you must pass all default bounding boxes corners coordinates!')
            xmin, ymin, xmax, ymax = (f32(v) for v in corners)
            cx, cy, w, h = self._coordinates_corners_to_centroids(xmin, ymin, xmax, ymax)
        elif all(v is None for v in corners):
            if any(v is None for v in centroids):
                raise ValueError('you must pass all default bounding boxes centroids coordinates!')
            cx, cy, w, h = (f32(v) for v in centroids)
            xmin, ymin, xmax, ymax = self._coordinates_centroids_to_corners(cx, cy, w, h)
        else:
            # passing both sets is rejected, exactly like the reference (its "both" branch is unreachable: quirk Q5)
            raise ValueError('you must pass all default bounding boxes centroids coordinates, or corners coordinates or both!')
        self.xmin_boxes_default, self.ymin_boxes_default, self.xmax_boxes_default, self.ymax_boxes_default = xmin, ymin, xmax, ymax
        self.center_x_boxes_default, self.center_y_boxes_default, self.width_boxes_default, self.height_boxes_default = cx, cy, w, h
        self.boxes_area_default = ((ymax - ymin + 1.0) * (xmax - xmin + 1.0))[:, None]
        self.augmentation_horizontal_flip = augmentation_horizontal_flip
        self._rng = np.random.default_rng(1993)
        self._dev = None

    # ---- coordinate helpers (reference datacoder.py:119-175)
    @staticmethod
    def _coordinates_corners_to_centroids(xmin, ymin, xmax, ymax):
        return (xmax + xmin) / 2.0, (ymax + ymin) / 2.0, xmax - xmin + 1.0, ymax - ymin + 1.0

    @staticmethod
    def _coordinates_centroids_to_corners(center_x, center_y, width, height):
        return center_x - (width - 1.0) / 2.0, center_y - (height - 1.0) / 2.0, center_x + (width - 1.0) / 2.0, center_y + (height - 1.0) / 2.0

    @property
    def _stds(self):
        return (self.standard_deviation_center_x_offsets, self.standard_deviation_center_y_offsets,
                self.standard_deviation_width_offsets, self.standard_deviation_height_offsets)

    # ---- GPU encode
    def _device_anchors(self):
        if self._dev is None:
            from . import _engine
            ctx = _engine.default_context()
            corners = np.stack([self.xmin_boxes_default, self.ymin_boxes_default, self.xmax_boxes_default, self.ymax_boxes_default], axis=1)
            self._dev = (ctx, ctx.array(corners.astype(np.float32)))
        return self._dev

    def encode_batch(self, labels_boxes: Sequence[np.ndarray], to_host: bool = True):
        """labels_boxes: per image an array (G, 5) = (label, xmin, ymin, xmax, ymax).
        -> labels (B, A, num_classes) one-hot, boxes (B, A, 4) offsets (NumPy, or DeviceBuffers with to_host=False)."""
        ctx, anchors = self._device_anchors()
        b = len(labels_boxes)
        gmax = max(1, max((np.asarray(g).reshape(-1, 5).shape[0] for g in labels_boxes), default=1))
        gt = np.zeros((b, gmax, 5), np.float32)
        cnt = np.zeros(b, np.int32)
        for i, g in enumerate(labels_boxes):
            g = np.asarray(g, np.float32).reshape(-1, 5)
            gt[i, :g.shape[0]] = g
            cnt[i] = g.shape[0]
        a = anchors.shape[0]
        labels, boxes = ctx.empty((b, a, self.num_classes)), ctx.empty((b, a, 4))
        ctx.call("ssdseg_encode_targets", anchors, a, ctx.array(gt), ctx.array(cnt), b, gmax, self.num_classes, float(self.iou_threshold),
                 (C.c_float * 4)(*self._stds), labels, boxes, None)
        if to_host:
            return labels.download(), boxes.download()
        return labels, boxes

    def _flip_boxes(self, gt: np.ndarray) -> np.ndarray:
        """x -> W - x, with W the image width, NOT W-1 (quirk Q8, reference datacoder.py:203)"""
        out = gt.copy()
        out[:, 1], out[:, 3] = self.image_width - gt[:, 3], self.image_width - gt[:, 1]
        return out

    def _encode_ground_truth_labels_boxes(self, path_file_labels_boxes: str, augment_with_horizontal_flip: bool):
        """CSV rows `label,xmin,ymin,xmax,ymax` -> (labels (A, C), boxes (A, 4)) (reference datacoder.py:177-300)."""
        with open(path_file_labels_boxes, "r", newline="") as f:
            text = f.read().strip()
        rows = [r for r in text.replace("\r\n", "\n").split("\n") if r]
        gt = np.array([[float(v) for v in r.split(",")] for r in rows], np.float32).reshape(-1, 5)
        if augment_with_horizontal_flip:
            gt = self._flip_boxes(gt)
        labels, boxes = self.encode_batch([gt])
        return labels[0], boxes[0]

    def read_and_encode(self, path_file_image: str, path_file_mask: str, path_file_labels_boxes: str):
        """(image, {'output-mask', 'output-labels', 'output-boxes'}) for one sample (reference datacoder.py:302-347)."""
        image = read_image(path_file_image)
        from PIL import Image
        mask_idx = np.asarray(Image.open(path_file_mask).convert("L"), np.int64)
        mask = np.eye(self.num_classes, dtype=np.float32)[np.clip(mask_idx, 0, self.num_classes - 1)]
        mask[mask_idx >= self.num_classes] = 0.0        # tf.one_hot gives an all-zero row for out-of-range indices
        flip = bool(self.augmentation_horizontal_flip and self._rng.uniform(0, 1) >= 0.5)
        if flip:
            image, mask = image[:, ::-1].copy(), mask[:, ::-1].copy()
        labels, boxes = self._encode_ground_truth_labels_boxes(path_file_labels_boxes, flip)
        return image, {'output-mask': mask, 'output-labels': labels, 'output-boxes': boxes}

    # ---- compact hand-over: what the files hold goes to the GPU, the expansion happens there (csrc/inputs.hip)
    def read_compact(self, path_file_image: str, path_file_mask: str, path_file_labels_boxes: str):
        """one sample as (uint8 image (H, W, 3), uint8 class-index mask (H, W), ground truth (G, 5), flip flag): the inputs of
        read_and_encode before its float expansion (reference datacoder.py:325-345); `compact_batch` stacks them"""
        from PIL import Image
        image = np.asarray(Image.open(path_file_image).convert("RGB"), np.uint8)
        mask_idx = np.asarray(Image.open(path_file_mask).convert("L"), np.uint8)
        with open(path_file_labels_boxes, "r", newline="") as f:
            text = f.read().strip()
        rows = [r for r in text.replace("\r\n", "\n").split("\n") if r]
        gt = np.array([[float(v) for v in r.split(",")] for r in rows], np.float32).reshape(-1, 5)
        flip = bool(self.augmentation_horizontal_flip and self._rng.uniform(0, 1) >= 0.5)
        return image, mask_idx, gt, flip

    def compact_batch(self, samples) -> "CompactBatch":
        images, masks, gts, flips = zip(*samples)
        return CompactBatch(np.stack(images), np.stack(masks), list(gts), np.asarray(flips, np.uint8), self)

    # ---- decode of GROUND-TRUTH offsets (reference datacoder.py:349-432)
    def decode_to_centroids(self, offsets_centroids, output_decoded_centroids_separately: bool = False):
        o = np.asarray(offsets_centroids, np.float32)
        sx, sy, sw, sh = (np.float32(s) for s in self._stds)
        not_background = (np.abs(o).sum(axis=-1) > 0.0).astype(np.float32)
        center_x = (o[:, 0] * sx * self.width_boxes_default + self.center_x_boxes_default) * not_background
        center_y = (o[:, 1] * sy * self.height_boxes_default + self.center_y_boxes_default) * not_background
        width = (np.exp(o[:, 2] * sw) - 1.0).astype(np.float32) * self.width_boxes_default * not_background
        height = (np.exp(o[:, 3] * sh) - 1.0).astype(np.float32) * self.height_boxes_default * not_background
        if output_decoded_centroids_separately:
            return center_x, center_y, width, height
        return np.stack([center_x, center_y, width, height], axis=1)

    def decode_to_corners(self, offsets_centroids, output_decoded_corners_separately: bool = False):
        center_x, center_y, width, height = self.decode_to_centroids(offsets_centroids, True)
        xmin, ymin, xmax, ymax = self._coordinates_centroids_to_corners(center_x, center_y, width, height)
        not_background = ((np.abs(center_x) + np.abs(center_y) + np.abs(width) + np.abs(height)) > 0.0).astype(np.float32)
        xmin, ymin, xmax, ymax = xmin * not_background, ymin * not_background, xmax * not_background, ymax * not_background
        if output_decoded_corners_separately:
            return xmin, ymin, xmax, ymax
        return np.stack([xmin, ymin, xmax, ymax], axis=1)


class CompactBatch:
    """A training batch as the files hold it: uint8 pixels (B, H, W, 3), uint8 class indices (B, H, W), per-sample ground-truth
    rows (label, xmin, ymin, xmax, ymax) and flip flags -- 39 MB instead of the 285 MB of float32 tensors at batch 32, 480x640.
    `Model.fit` / `train_on_batch` accept it in place of (images, targets): float conversion, one-hot, mirroring and the anchor
    encoding run on the GPU (ssdseg_expand_inputs, ssdseg_flip_gt_boxes, ssdseg_encode_targets)."""

    def __init__(self, images_u8, mask_index_u8, ground_truth, flip, encoder: "DataEncoderDecoder"):
        self.images = np.ascontiguousarray(images_u8, np.uint8)
        self.mask_index = np.ascontiguousarray(mask_index_u8, np.uint8)
        if self.images.ndim != 4 or self.images.shape[-1] != 3 or self.mask_index.shape != self.images.shape[:3]:
            raise ValueError(f"compact batch: images {self.images.shape} must be (B, H, W, 3) and masks {self.mask_index.shape} (B, H, W)")
        self.ground_truth = [np.asarray(g, np.float32).reshape(-1, 5) for g in ground_truth]
        if len(self.ground_truth) != self.images.shape[0]:
            raise ValueError("compact batch: one ground-truth array per image")
        self.flip = None if flip is None else np.ascontiguousarray(flip, np.uint8).reshape(self.images.shape[0])
        self.encoder = encoder

    def __len__(self):
        return self.images.shape[0]


_aug_rng = np.random.default_rng(1993)


def _rgb_to_hsv(rgb):
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    mx, mn = rgb.max(-1), rgb.min(-1)
    d = mx - mn
    s = np.where(mx > 0, d / np.where(mx > 0, mx, 1), 0)
    dd = np.where(d > 0, d, 1)
    h = np.where(mx == r, (g - b) / dd, np.where(mx == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd))
    h = np.where(d > 0, (h / 6.0) % 1.0, 0.0)
    return np.stack([h, s, mx], -1)


def _hsv_to_rgb(hsv):
    h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    dh = h * 6.0
    k = lambda n: (n + dh) % 6.0
    f = lambda n: v - v * s * np.clip(np.minimum(k(n), 4.0 - k(n)), 0.0, 1.0)
    return np.stack([f(5), f(3), f(1)], -1)


def _augment_rgb(x, hue_delta, saturation_factor, contrast_factor, brightness_delta):
    """the four adjustments with GIVEN random draws (tf.image.adjust_hue / adjust_saturation / adjust_contrast /
    adjust_brightness on float images, then the clip of reference datacoder.py:464)"""
    x = np.asarray(x, np.float32)
    hsv = _rgb_to_hsv(x)
    hsv[..., 0] = (hsv[..., 0] + hue_delta) % 1.0
    x = _hsv_to_rgb(hsv)
    hsv = _rgb_to_hsv(x)
    hsv[..., 1] = np.clip(hsv[..., 1] * saturation_factor, 0.0, 1.0)
    x = _hsv_to_rgb(hsv)
    mean = x.mean(axis=(1, 2), keepdims=True)          # per image and channel
    x = (x - mean) * contrast_factor + mean
    x = x + brightness_delta
    return np.clip(x, 0.0, 255.0).astype(np.float32)


def augmentation_rgb_channels(image_batch, targets_batch):
    """random hue (+-0.05), saturation (0.95..1.05), contrast (0.9..1.1), brightness (+-0.10) then clip to [0, 255]
    (reference datacoder.py:452-464; the deltas are the [0,1]-scale ones applied to 0..255 images, quirk Q11).
    Host-side input-pipeline step; TF's exact RNG streams are not reproduced."""
    draws = (_aug_rng.uniform(-0.05, 0.05), _aug_rng.uniform(0.95, 1.05), _aug_rng.uniform(0.90, 1.10), _aug_rng.uniform(-0.10, 0.10))
    return _augment_rgb(image_batch, *draws), targets_batch


def read_image(path_file_image: str) -> np.ndarray:
    """PNG -> float32 (H, W, 3) in 0..255 (reference datacoder.py:468-484)."""
    from PIL import Image
    return np.asarray(Image.open(path_file_image).convert("RGB"), np.float32)
