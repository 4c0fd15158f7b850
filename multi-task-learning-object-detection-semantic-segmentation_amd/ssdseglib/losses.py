"""Training losses with the reference's API (reference losses.py): every function maps (y_true, y_pred) to one loss
value per batch item, shape (batch,).  Called directly they run the HIP kernels on cuda:0 and return NumPy arrays;
passed to `model.compile(loss={...})` they select the fused loss ops of the engine (`_engine.configure_losses`),
where the gradient comes out of the same launch.
"""
import ctypes as C
from typing import Callable, List

import numpy as np


def _ctx():
    from . import _engine
    return _engine.default_context()


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def localization_loss(y_true, y_pred):
    """smooth-L1 over non-background anchors / max(#non-background, 1) (reference losses.py:21-49)."""
    y_true, y_pred = _f32(y_true), _f32(y_pred)
    b, a, _ = y_true.shape
    ctx = _ctx()
    dummy_labels = np.zeros((b, a, 4), np.float32)
    dummy_labels[..., 0] = 1.0
    loc = ctx.empty(b)
    ctx.call("ssdseg_det_loss", ctx.array(dummy_labels), ctx.array(np.full((b, a, 4), 0.25, np.float32)), ctx.array(y_true), ctx.array(y_pred),
             b, a, 4, 1.0, None, loc, None, None, None)
    return loc.download()


def confidence_loss(y_true, y_pred):
    """softmax cross-entropy with batch-global 3:1 hard-negative mining (reference losses.py:70-172);
    y_pred are probabilities."""
    y_true, y_pred = _f32(y_true), _f32(y_pred)
    b, a, c = y_true.shape
    ctx = _ctx()
    zeros = ctx.zeros((b, a, 4))
    conf = ctx.empty(b)
    ctx.call("ssdseg_det_loss", ctx.array(y_true), ctx.array(y_pred), zeros, zeros, b, a, c, 1.0, conf, None, None, None, None)
    return conf.download()


def _mask_loss(mode: int, name: str, kind: str, classes_weights: List[float]) -> Callable:
    weights = tuple(float(w) for w in classes_weights)

    def loss_fn(y_true, y_pred):
        y_true, y_pred = _f32(y_true), _f32(y_pred)
        n, c = y_true.shape[0], y_true.shape[-1]
        hw = int(np.prod(y_true.shape[1:-1]))
        ctx = _ctx()
        out = ctx.empty(n)
        ctx.call("ssdseg_dice_loss", ctx.array(y_true), ctx.array(y_pred), n, hw, c, (C.c_float * 4)(*weights), mode, out)
        return out.download()

    loss_fn.__name__ = name
    loss_fn.classes_weights = weights
    loss_fn.loss_kind = kind
    return loss_fn


def dice(classes_weights: List[float]) -> Callable:
    """weighted dice loss on probabilities (reference losses.py:204-216)."""
    return _mask_loss(0, "dice_loss", "dice", classes_weights)


def dice_square(classes_weights: List[float]) -> Callable:
    """weighted squared-denominator dice loss (reference losses.py:250-262)."""
    return _mask_loss(1, "dice_square_loss", "dice_square", classes_weights)


def cross_entropy(classes_weights: List[float]) -> Callable:
    """weighted pixel cross-entropy on probabilities, summed over pixels and classes (reference losses.py:294-305);
    the loss NB03#cell10 trains the mask head with."""
    return _mask_loss(2, "cross_entropy_loss", "cross_entropy", classes_weights)
