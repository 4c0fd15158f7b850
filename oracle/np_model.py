"""CPU oracle: layer-by-layer NumPy interpreter of a `ssdseglib._graph.Model` -- TEST INFRASTRUCTURE, NOT PRODUCT.

Executes the graph the way Keras would (one unfused op per layer, training-mode BatchNormalization), forward
and backward, using the restated TF semantics in `np_ops.py`.  It shares only the *graph description* with the
product (layer names / shapes / weights, pinned by the reference's own model.summary() output); none of the product's
fusion, lowering or kernels.  Same pinning status as np_ops.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from . import np_ops as O


def _layer_kind(layer) -> str:
    return type(layer).__name__


class NpModel:
    def __init__(self, model, dtype=np.float32):
        self.model = model
        self.dt = dtype
        self.weights: Dict[str, Dict[str, np.ndarray]] = {
            l.name: {k: v.astype(dtype) for k, v in l.weights.items()} for l in model.layers if l.weights}
        self.vals: Dict[int, np.ndarray] = {}
        self.cache: Dict[str, dict] = {}

    def set_weights_from(self, getter):
        """getter(layer) -> list of arrays in Keras order"""
        for l in self.model.layers:
            if l.weights:
                for k, a in zip(l.weights, getter(l)):
                    self.weights[l.name][k] = np.asarray(a, self.dt)

    # ------------------------------------------------------------------ forward
    def forward(self, x, training=True) -> List[np.ndarray]:
        self.vals, self.cache = {}, {}
        self.training = training
        for l in self.model.layers:
            ins = [self.vals[id(t)] for t in l.inbound]
            out = self._fwd(l, ins, x)
            if isinstance(out, list):
                for t, o in zip(l.outputs, out):
                    self.vals[id(t)] = o
            else:
                self.vals[id(l.outputs[0])] = out
        return [self.vals[id(t)] for t in self.model.outputs]

    def value(self, layer_name: str) -> np.ndarray:
        return self.vals[id(self.model.get_layer(layer_name).outputs[0])]

    def _fwd(self, l, ins, x):
        k = _layer_kind(l)
        w = self.weights.get(l.name, {})
        dt = self.dt
        if k == "InputLayer":
            return np.asarray(x, dt)
        if k == "Rescaling":
            return O.rescale(ins[0], l.scale, l.offset)
        if k == "Conv2D":
            return O.conv2d_fwd(ins[0], w["kernel"], l.strides[0], l.dilation_rate[0], w.get("bias"))
        if k == "DepthwiseConv2D":
            return O.dwconv_fwd(ins[0], w["depthwise_kernel"][..., 0], l.strides[0], l.dilation_rate[0])
        if k == "SeparableConv2D":
            mid = O.dwconv_fwd(ins[0], w["depthwise_kernel"][..., 0], l.strides[0], l.dilation_rate[0])
            self.cache[l.name] = dict(mid=mid)
            return O.conv2d_fwd(mid, w["pointwise_kernel"])
        if k == "BatchNormalization":
            if self.training:
                z, cache = O.bn_train_fwd(ins[0], w["gamma"], w["beta"], l.epsilon)
                self.cache[l.name] = cache
                return z
            sc, sh = O.bn_infer_affine(w["gamma"], w["beta"], w["moving_mean"], w["moving_variance"], l.epsilon)
            return ins[0] * sc + sh
        if k == "ReLU":
            act = O.ACT_RELU if l.max_value is None else (O.ACT_ZERO if l.max_value == 0.0 else O.ACT_RELU6)
            self.cache[l.name] = dict(act=act)
            return O.act_fwd(ins[0], act)
        if k == "Add":
            return ins[0] + ins[1]
        if k == "Concatenate":
            return np.concatenate(ins, axis=l.axis_resolved)
        if k == "GlobalAveragePooling2D":
            return O.gap_fwd(ins[0])
        if k == "UpSampling2D":
            return O.bilinear_fwd(ins[0], l.size[0], l.size[1])
        if k == "MaxPooling2D":
            return O.maxpool3x3s2_fwd(ins[0])
        if k == "Softmax":
            return O.softmax(ins[0])
        if k == "Reshape":
            return ins[0].reshape((ins[0].shape[0],) + tuple(l.outputs[0].shape[1:]))
        if k == "Permute":
            return ins[0].transpose((0,) + tuple(l.dims))
        if k == "Split":
            return list(np.split(ins[0], len(l.outputs), axis=l.axis))
        if k == "DecodeBoxesCentroidsOffsets":
            cent = np.stack([l.center_x_boxes_default, l.center_y_boxes_default, l.width_boxes_default, l.height_boxes_default], axis=1)
            stds = (l.standard_deviation_center_x_offsets, l.standard_deviation_center_y_offsets,
                    l.standard_deviation_width_offsets, l.standard_deviation_height_offsets)
            return O.decode_to_corners_pred(ins[0], cent, stds)
        if k == "SegmentationSuppression":
            return O.seg_suppress(ins[0], ins[1])
        if k == "NonMaximumSuppression":
            out, valid = O.combined_nms(ins[0], ins[1], l.max_number_of_boxes_per_class, l.max_number_of_boxes_per_sample,
                                        l.boxes_iou_threshold, l.labels_probability_threshold)
            self.cache[l.name] = dict(valid=valid)
            return out
        raise NotImplementedError(k)

    # ------------------------------------------------------------------ backward
    def backward(self, output_grads: List[Optional[np.ndarray]], relu_masks: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, Dict[str, np.ndarray]]:
        """output_grads[i] = dL/d(model.outputs[i]) -> {layer name: {weight name: gradient}}.

        relu_masks (optional): {ReLU layer name: 0/1 derivative mask}.  ReLU6's derivative is discontinuous, so in a
        deep fp32 network a handful of pre-activations within rounding distance of 0 or 6 get a different mask on any two
        implementations (fp32 NumPy vs fp64 NumPy differ the same way).  Whole-network gradient comparisons therefore
        pass the masks observed on the device; the mask arithmetic itself is pinned by the per-kernel tests."""
        relu_masks = relu_masks or {}
        g: Dict[int, np.ndarray] = {}

        def acc(t, v):
            if v is None:
                return
            g[id(t)] = v if id(t) not in g else g[id(t)] + v

        for t, og in zip(self.model.outputs, output_grads):
            acc(t, None if og is None else np.asarray(og, self.dt))
        grads: Dict[str, Dict[str, np.ndarray]] = {}
        for l in reversed(self.model.layers):
            k = _layer_kind(l)
            if k == "InputLayer":
                continue
            outs = [g.get(id(t)) for t in l.outputs]
            if all(o is None for o in outs):
                continue
            ins = [self.vals[id(t)] for t in l.inbound]
            w = self.weights.get(l.name, {})
            go = outs[0]
            if k == "Rescaling":
                pass
            elif k == "Conv2D":
                dx, dw, db = O.conv2d_bwd(ins[0], w["kernel"], go, l.strides[0], l.dilation_rate[0])
                grads[l.name] = {"kernel": dw}
                if "bias" in w:
                    grads[l.name]["bias"] = db
                acc(l.inbound[0], dx)
            elif k == "DepthwiseConv2D":
                dx, dw = O.dwconv_bwd(ins[0], w["depthwise_kernel"][..., 0], go, l.strides[0], l.dilation_rate[0])
                grads[l.name] = {"depthwise_kernel": dw[..., None]}
                acc(l.inbound[0], dx)
            elif k == "SeparableConv2D":
                mid = self.cache[l.name]["mid"]
                dmid, dpw, _ = O.conv2d_bwd(mid, w["pointwise_kernel"], go)
                dx, ddw = O.dwconv_bwd(ins[0], w["depthwise_kernel"][..., 0], dmid, l.strides[0], l.dilation_rate[0])
                grads[l.name] = {"depthwise_kernel": ddw[..., None], "pointwise_kernel": dpw}
                acc(l.inbound[0], dx)
            elif k == "BatchNormalization":
                dy, dgamma, dbeta = O.bn_train_bwd(go, ins[0], w["gamma"], self.cache[l.name])
                grads[l.name] = {"gamma": dgamma, "beta": dbeta}
                acc(l.inbound[0], dy)
            elif k == "ReLU":
                mask = relu_masks.get(l.name)
                acc(l.inbound[0], go * (O.act_mask(ins[0], self.cache[l.name]["act"]) if mask is None else mask.astype(go.dtype)))
            elif k == "Add":
                acc(l.inbound[0], go)
                acc(l.inbound[1], go)
            elif k == "Concatenate":
                off = 0
                ax = l.axis_resolved
                for t in l.inbound:
                    c = t.shape[ax]
                    sl = [slice(None)] * go.ndim
                    sl[ax] = slice(off, off + c)
                    acc(t, go[tuple(sl)])
                    off += c
            elif k == "GlobalAveragePooling2D":
                acc(l.inbound[0], O.gap_bwd(go, ins[0].shape[1], ins[0].shape[2]))
            elif k == "UpSampling2D":
                acc(l.inbound[0], O.bilinear_bwd(go, l.size[0], l.size[1]))
            elif k == "Softmax":
                acc(l.inbound[0], O.softmax_bwd(self.vals[id(l.outputs[0])], go))
            elif k == "Reshape":
                acc(l.inbound[0], go.reshape(ins[0].shape))
            elif k == "Permute":
                inv = np.argsort((0,) + tuple(l.dims))
                acc(l.inbound[0], go.transpose(inv))
            elif k == "Split":
                parts = [o if o is not None else np.zeros_like(self.vals[id(t)]) for o, t in zip(outs, l.outputs)]
                acc(l.inbound[0], np.concatenate(parts, axis=l.axis))
            elif k == "MaxPooling2D":
                acc(l.inbound[0], O.maxpool3x3s2_bwd(ins[0], go))
            else:
                raise NotImplementedError(k)
        return grads
