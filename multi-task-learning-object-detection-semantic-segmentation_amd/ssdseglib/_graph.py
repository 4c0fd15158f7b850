"""A minimal functional layer graph with the Keras vocabulary the reference's builders use.

The reference builds its network with `tf.keras.layers.*` functional calls (models.py:47-343, blocks.py:4-157) and
drives it through `tf.keras.Model` (NB03#cell12-25).  Keras itself is third-party and absent here; this module
supplies just the *graph description* those builders need -- symbolic tensors, named layers with Keras' weight
order/shapes/initialisers, topological model -- so that `ssdseglib.models` reads like the reference and
`model.summary()` reproduces NB03#cell12's table.  No arithmetic happens here: `_engine.py` lowers the graph to
fused HIP launches.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

_rng = np.random.default_rng(1993)  # NB03#cell2 seeds with 1993


def set_seed(seed: int) -> None:
    global _rng
    _rng = np.random.default_rng(seed)


def _pair(v) -> Tuple[int, int]:
    return (int(v), int(v)) if isinstance(v, (int, np.integer)) else (int(v[0]), int(v[1]))


def _same_out(size: int, stride: int) -> int:
    return -(-size // stride)


def _glorot_uniform(shape: Sequence[int]) -> np.ndarray:
    """Keras default kernel initialiser: fan_in/fan_out with the receptive field folded in."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return _rng.uniform(-limit, limit, size=shape).astype(np.float32)


class KTensor:
    """Symbolic tensor: shape with a None batch axis, the layer that produced it and which of its outputs."""

    def __init__(self, shape, layer: "Layer", index: int = 0):
        self.shape = tuple(shape)
        self.layer = layer
        self.index = index

    @property
    def name(self) -> str:
        return self.layer.name

    def get_shape(self):
        return _Shape(self.shape)

    def __repr__(self):
        return f"<KTensor {self.layer.name}[{self.index}] {self.shape}>"


class _Shape(tuple):
    def as_list(self):
        return list(self)


class Layer:
    """Base class.  Subclasses set `weights` (dict name -> ndarray, in Keras order) and `trainable_names`."""
    type_name = "Layer"
    _uid: Dict[str, int] = {}

    def __init__(self, name: Optional[str] = None, **kwargs):
        if name is None:
            base = self.type_name.lower()
            Layer._uid[base] = Layer._uid.get(base, 0) + 1
            name = f"{base}_{Layer._uid[base]}"
        self.name = name
        self.trainable = True
        self.inbound: List[KTensor] = []
        self.outputs: List[KTensor] = []
        self.weights: Dict[str, np.ndarray] = {}
        self.trainable_names: Tuple[str, ...] = ()
        self._engine_sync = None  # set by the engine: (pull weights from device, push weights to device)

    # -- functional call
    def __call__(self, *args, **kwargs):
        ins = []
        for a in args:
            ins += list(a) if isinstance(a, (list, tuple)) else [a]
        ins = [a for a in ins if isinstance(a, KTensor)]
        ins += [v for v in kwargs.values() if isinstance(v, KTensor)]
        if self.inbound:
            raise ValueError(f"layer {self.name} is already connected (layer sharing is not supported)")
        self.inbound = ins
        shapes = self.build([t.shape for t in ins])
        if isinstance(shapes, list):
            self.outputs = [KTensor(s, self, i) for i, s in enumerate(shapes)]
            return list(self.outputs)
        self.outputs = [KTensor(shapes, self, 0)]
        return self.outputs[0]

    def build(self, input_shapes):
        return input_shapes[0]

    @property
    def output(self):
        return self.outputs[0] if len(self.outputs) == 1 else list(self.outputs)

    @property
    def output_shape(self):
        return self.outputs[0].shape

    def count_params(self) -> int:
        return int(sum(w.size for w in self.weights.values()))

    def count_trainable(self) -> int:
        return int(sum(self.weights[n].size for n in self.trainable_names)) if self.trainable else 0

    def get_weights(self) -> List[np.ndarray]:
        if self._engine_sync is not None:
            self._engine_sync[0](self)
        return [w.copy() for w in self.weights.values()]

    def set_weights(self, values: Sequence[np.ndarray]) -> None:
        names = list(self.weights)
        if len(values) != len(names):
            raise ValueError(f"layer {self.name} expects {len(names)} weight arrays, got {len(values)}")
        for n, v in zip(names, values):
            v = np.asarray(v, dtype=np.float32)
            if v.shape != self.weights[n].shape:
                raise ValueError(f"layer {self.name} weight {n}: shape {v.shape} != {self.weights[n].shape}")
            self.weights[n] = v.copy()
        if self._engine_sync is not None:
            self._engine_sync[1](self)

    def get_config(self) -> dict:
        return {"name": self.name}


class InputLayer(Layer):
    type_name = "InputLayer"

    def __init__(self, shape, dtype="float32", name=None):
        super().__init__(name=name)
        self.outputs = [KTensor((None,) + tuple(shape), self, 0)]

    @property
    def output_shape(self):
        return [self.outputs[0].shape]  # Keras prints a list for InputLayer


def Input(shape, dtype="float32", name=None) -> KTensor:
    return InputLayer(shape, dtype, name).outputs[0]


class Rescaling(Layer):
    type_name = "Rescaling"

    def __init__(self, scale, offset=0.0, name=None):
        super().__init__(name=name)
        self.scale, self.offset = float(scale), float(offset)


class Conv2D(Layer):
    type_name = "Conv2D"

    def __init__(self, filters, kernel_size, strides=1, padding="same", dilation_rate=1, use_bias=True, name=None):
        super().__init__(name=name)
        assert padding == "same", "only SAME padding is used by ssdseglib"
        self.filters, self.kernel_size, self.strides = int(filters), _pair(kernel_size), _pair(strides)
        self.dilation_rate, self.use_bias = _pair(dilation_rate), bool(use_bias)

    def build(self, input_shapes):
        n, h, w, c = input_shapes[0]
        self.weights = {"kernel": _glorot_uniform(self.kernel_size + (c, self.filters))}
        self.trainable_names = ("kernel",)
        if self.use_bias:
            self.weights["bias"] = np.zeros(self.filters, np.float32)
            self.trainable_names = ("kernel", "bias")
        return (n, _same_out(h, self.strides[0]), _same_out(w, self.strides[1]), self.filters)


class DepthwiseConv2D(Layer):
    type_name = "DepthwiseConv2D"

    def __init__(self, kernel_size, strides=1, padding="same", depth_multiplier=1, dilation_rate=1, use_bias=True, name=None):
        super().__init__(name=name)
        assert padding == "same" and depth_multiplier == 1 and not use_bias
        self.kernel_size, self.strides, self.dilation_rate = _pair(kernel_size), _pair(strides), _pair(dilation_rate)

    def build(self, input_shapes):
        n, h, w, c = input_shapes[0]
        self.weights = {"depthwise_kernel": _glorot_uniform(self.kernel_size + (c, 1))}
        self.trainable_names = ("depthwise_kernel",)
        return (n, _same_out(h, self.strides[0]), _same_out(w, self.strides[1]), c)


class SeparableConv2D(Layer):
    type_name = "SeparableConv2D"

    def __init__(self, filters, kernel_size, strides=1, padding="same", dilation_rate=1, depth_multiplier=1, use_bias=True, name=None):
        super().__init__(name=name)
        assert padding == "same" and depth_multiplier == 1 and not use_bias
        self.filters, self.kernel_size, self.strides = int(filters), _pair(kernel_size), _pair(strides)
        self.dilation_rate = _pair(dilation_rate)

    def build(self, input_shapes):
        n, h, w, c = input_shapes[0]
        self.weights = {"depthwise_kernel": _glorot_uniform(self.kernel_size + (c, 1)),
                        "pointwise_kernel": _glorot_uniform((1, 1, c, self.filters))}
        self.trainable_names = ("depthwise_kernel", "pointwise_kernel")
        return (n, _same_out(h, self.strides[0]), _same_out(w, self.strides[1]), self.filters)


class BatchNormalization(Layer):
    type_name = "BatchNormalization"

    def __init__(self, momentum=0.99, epsilon=1e-3, name=None):
        super().__init__(name=name)
        self.momentum, self.epsilon = float(momentum), float(epsilon)

    def build(self, input_shapes):
        c = input_shapes[0][-1]
        self.weights = {"gamma": np.ones(c, np.float32), "beta": np.zeros(c, np.float32),
                        "moving_mean": np.zeros(c, np.float32), "moving_variance": np.ones(c, np.float32)}
        self.trainable_names = ("gamma", "beta")
        return input_shapes[0]


class ReLU(Layer):
    type_name = "ReLU"

    def __init__(self, max_value=None, name=None):
        super().__init__(name=name)
        self.max_value = None if max_value is None else float(max_value)


class Add(Layer):
    type_name = "Add"


class Concatenate(Layer):
    type_name = "Concatenate"

    def __init__(self, axis=-1, name=None):
        super().__init__(name=name)
        self.axis = axis

    def build(self, input_shapes):
        rank = len(input_shapes[0])
        ax = self.axis % rank
        self.axis_resolved = ax
        out = list(input_shapes[0])
        out[ax] = sum(s[ax] for s in input_shapes)
        return tuple(out)


class GlobalAveragePooling2D(Layer):
    type_name = "GlobalAveragePooling2D"

    def __init__(self, data_format="channels_last", keepdims=False, name=None):
        super().__init__(name=name)
        assert keepdims and data_format == "channels_last"

    def build(self, input_shapes):
        n, h, w, c = input_shapes[0]
        return (n, 1, 1, c)


class UpSampling2D(Layer):
    type_name = "UpSampling2D"

    def __init__(self, size=(2, 2), interpolation="nearest", name=None):
        super().__init__(name=name)
        assert interpolation == "bilinear"
        self.size = _pair(size)

    def build(self, input_shapes):
        n, h, w, c = input_shapes[0]
        return (n, h * self.size[0], w * self.size[1], c)


class MaxPooling2D(Layer):
    type_name = "MaxPooling2D"

    def __init__(self, pool_size=2, strides=None, padding="valid", name=None):
        super().__init__(name=name)
        self.pool_size, self.strides = _pair(pool_size), _pair(strides if strides is not None else pool_size)
        assert padding == "same" and self.pool_size == (3, 3) and self.strides == (2, 2)

    def build(self, input_shapes):
        n, h, w, c = input_shapes[0]
        return (n, _same_out(h, 2), _same_out(w, 2), c)


class Softmax(Layer):
    type_name = "Softmax"


class Reshape(Layer):
    type_name = "Reshape"

    def __init__(self, target_shape, name=None):
        super().__init__(name=name)
        self.target_shape = tuple(target_shape)

    def build(self, input_shapes):
        total = int(np.prod(input_shapes[0][1:]))
        tgt = list(self.target_shape)
        if -1 in tgt:
            known = int(np.prod([t for t in tgt if t != -1]))
            tgt[tgt.index(-1)] = total // known
        assert int(np.prod(tgt)) == total, f"{self.name}: cannot reshape {input_shapes[0]} to {self.target_shape}"
        return (None,) + tuple(tgt)


class Permute(Layer):
    type_name = "Permute"

    def __init__(self, dims, name=None):
        super().__init__(name=name)
        self.dims = tuple(dims)

    def build(self, input_shapes):
        s = input_shapes[0]
        return (None,) + tuple(s[d] for d in self.dims)


# ---------------------------------------------------------------------------------------------------------- model
class Model:
    """Topologically ordered layer graph between `inputs` and `outputs` (the Keras subset NB03 uses)."""

    def __init__(self, inputs, outputs, name="model"):
        self.name = name
        self.inputs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        self.outputs = list(outputs) if isinstance(outputs, (list, tuple)) else [outputs]
        self.layers: List[Layer] = self._toposort()
        self._by_name = {l.name: l for l in self.layers}
        if len(self._by_name) != len(self.layers):
            raise ValueError("duplicate layer names in model")
        self.output_names = [t.layer.name for t in self.outputs]
        self._engine = None
        self._compiled = None

    def _toposort(self) -> List[Layer]:
        """Keras orders layers by depth from the outputs (deepest first), ties in creation/visit order; the
        resulting sequence for the reference graph is the one NB03#cell12 prints."""
        first_seen: Dict[int, int] = {}      # pre-order index of the depth-first walk from the outputs
        finished: List[Layer] = []           # post-order: producers before consumers
        done = set()

        def walk(layer: Layer):
            if id(layer) in done:
                return
            if id(layer) not in first_seen:
                first_seen[id(layer)] = len(first_seen)
            for t in layer.inbound:
                walk(t.layer)
            done.add(id(layer))
            finished.append(layer)

        for t in self.outputs:
            walk(t.layer)
        for t in self.inputs:
            if id(t.layer) not in done:
                raise ValueError("graph disconnected: an input is not reachable from the outputs")
        # depth = longest path to an output, propagated consumers-first
        depth: Dict[int, int] = {id(l): 0 for l in finished}
        for layer in reversed(finished):
            for t in layer.inbound:
                depth[id(t.layer)] = max(depth[id(t.layer)], depth[id(layer)] + 1)
        return sorted(finished, key=lambda l: (-depth[id(l)], first_seen[id(l)]))

    def get_layer(self, name: str) -> Layer:
        if name not in self._by_name:
            raise ValueError(f"No such layer: {name}. Existing layers are: {list(self._by_name)[:5]}...")
        return self._by_name[name]

    def count_params(self) -> int:
        return sum(l.count_params() for l in self.layers)

    def summary(self, print_fn=print) -> None:
        total = self.count_params()
        trainable = sum(l.count_trainable() for l in self.layers)
        w = [29, 29, 10, 30]
        line = "_" * 98
        print_fn(f'Model: "{self.name}"')
        print_fn(line)
        print_fn(" " + "Layer (type)".ljust(w[0] - 1) + "Output Shape".ljust(w[1]) + "Param #".ljust(w[2]) + "Connected to")
        print_fn("=" * 98)
        for l in self.layers:
            shape = l.output_shape
            conn = [f"{t.layer.name}[0][{t.index}]" for t in l.inbound]
            print_fn(f" {l.name} ({l.type_name})  {shape}  {l.count_params()}  {conn}")
            print_fn("")
        print_fn("=" * 98)
        print_fn(f"Total params: {total} ({total * 4 / 2**20:.2f} MB)")
        print_fn(f"Trainable params: {trainable} ({trainable * 4 / 2**20:.2f} MB)")
        print_fn(f"Non-trainable params: {total - trainable} ({(total - trainable) * 4 / 2**10:.2f} KB)")
        print_fn(line)

    # ---- runtime surface: delegated to the HIP engine (lazy import keeps graph building GPU-free)
    def _get_engine(self, batch_size: int, training: bool):
        from . import _engine
        return _engine.engine_for(self, batch_size, training)

    def compile(self, optimizer=None, loss=None, loss_weights=None, metrics=None):
        self._compiled = dict(optimizer=optimizer, loss=loss or {}, loss_weights=loss_weights or {}, metrics=metrics or {})

    def __call__(self, x, training=False):
        from . import _engine
        return _engine.run_forward(self, x, training)

    def predict(self, data, verbose=0):
        from . import _engine
        return _engine.run_predict(self, data)

    def fit(self, data, epochs=1, validation_data=None, verbose=0):
        from . import _engine
        return _engine.run_fit(self, data, epochs, validation_data, verbose)

    def train_on_batch(self, x, y=None):
        from . import _engine
        return _engine.run_train_on_batch(self, x, y)

    def save(self, path):
        """Own checkpoint: one .npz keyed `<layer>/<weight>` with Keras' names/shapes."""
        blobs = {}
        for l in self.layers:
            for wname, arr in zip(l.weights, l.get_weights()):
                blobs[f"{l.name}/{wname}"] = arr
        np.savez(path if str(path).endswith(".npz") else str(path) + ".npz", **blobs)

    def load_weights(self, path):
        data = np.load(path if str(path).endswith(".npz") else str(path) + ".npz")
        for l in self.layers:
            if l.weights:
                l.set_weights([data[f"{l.name}/{w}"] for w in l.weights])


_creation_counter = [0]
_orig_layer_init = Layer.__init__


def _counting_init(self, *a, **k):
    _orig_layer_init(self, *a, **k)
    _creation_counter[0] += 1
    self._created = _creation_counter[0]


Layer.__init__ = _counting_init


def _creation_order(layers):
    return sorted(layers, key=lambda l: l._created)
