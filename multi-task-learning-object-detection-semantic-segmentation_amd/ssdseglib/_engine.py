"""Lowers a `_graph.Model` to a static list of fused HIP launches (forward list, reversed backward list).

This is the stand-in for the part of Keras the reference relies on (`Model.__call__/fit/predict`, GradientTape,
Adam -- third-party), written around the MI355X design:

  * conv -> BatchNormalization(train) -> ReLU6 is never three passes: the conv kernel writes the RAW tensor once
    plus per-block (sum, sumsq); `bn_finalize` makes per-channel (scale, shift); every consumer applies
    act(scale*y+shift) while loading (`Val` below is that lazy view).  Backward mirrors it: one reduction pass per
    BN gives (k1, k0) and the producing conv's backward kernels form dY on load (gradient view).
  * Concatenate over channels is a write offset (producers write straight into their slice of the concat
    buffer, `ld` = total channels), Reshape/Split are metadata.
  * everything stays resident in HBM (288 GB): no recomputation, no activation re-use planning.
  * the launch list is static (no Python-side decisions inside a step besides the op order).

Manual backward replaces autodiff: every op below carries its own `bwd`.  Gradient fan-in order is the fixed
reverse launch order (deterministic).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _graph as K
from . import _hip as H
from . import layers as L

ACT_NONE, ACT_RELU, ACT_RELU6, ACT_ZERO = H.ACT_NONE, H.ACT_RELU, H.ACT_RELU6, H.ACT_ZERO


def _nonneg(v: "Val") -> bool:
    """values known to be >= 0 (a ReLU applied on top of them is the identity)"""
    return v.act in (ACT_RELU, ACT_RELU6, ACT_ZERO) or bool(v.meta.get("nonneg"))


def _act_of(relu: K.ReLU) -> int:
    if relu.max_value is None:
        return ACT_RELU
    if relu.max_value == 0.0:
        return ACT_ZERO          # quirk Q1: Keras clips to [0, 0]
    if relu.max_value == 6.0:
        return ACT_RELU6
    raise NotImplementedError(f"ReLU(max_value={relu.max_value}) is not used by ssdseglib")


class Store:
    """A tensor region resident in HBM: rows m = n*h*w, c channels, row stride ld (>= c for concat slices)."""

    def __init__(self, eng: "Engine", n, h, w, c, name, buf: Optional[H.DeviceBuffer] = None, ld: Optional[int] = None,
                 parent: Optional["Store"] = None, coff: int = 0, need_grad: bool = True):
        self.eng, self.n, self.h, self.w, self.c, self.name = eng, n, h, w, c, name
        self.m = n * h * w
        self.ld = c if ld is None else ld
        self.parent, self.coff = parent, coff
        self.buf = buf if buf is not None else eng.ctx.empty((self.m, self.ld))
        self.need_grad = need_grad
        self._grad: Optional[H.DeviceBuffer] = None
        self.gwritten = False
        self.pending = None        # (buffer, ld): deferred residual contribution to this gradient (see defer_residual)
        self.split_parent = False  # channel Split views write disjoint ranges of this gradient: zero it, then everyone accumulates
        self.stats: Optional[H.DeviceBuffer] = None   # per-block (sum, sumsq) partials from the producing conv
        self.nparts = 0

    def slice(self, coff: int, c: int, name: str) -> "Store":
        assert coff + c <= self.c
        child = Store(self.eng, self.n, self.h, self.w, c, name, buf=self.buf.view(coff, (self.m * self.ld - coff,)), ld=self.ld,
                      parent=self, coff=coff, need_grad=self.need_grad)
        self.eng.stores.append(child)      # so its "gradient written" flag is reset with everyone else's each backward pass
        return child

    def _raw_grad(self) -> H.DeviceBuffer:
        if self._grad is None:
            if self.parent is not None:
                pg = self.parent._raw_grad()
                self._grad = pg.view(self.coff, (self.m * self.ld - self.coff,))
            else:
                self._grad = self.eng.ctx.zeros((self.m, self.ld))
        return self._grad

    @property
    def grad(self) -> H.DeviceBuffer:
        """the gradient buffer, complete as far as the backward pass has come (a deferred residual contribution is added first)"""
        self._materialise_pending()
        return self._raw_grad()

    def _written(self) -> bool:
        return self.gwritten or (self.parent is not None and self.parent.gwritten)

    def _materialise_pending(self):
        if self.pending is not None:
            buf, ld = self.pending
            self.pending = None
            self.eng.ctx.call("ssdseg_axpby", buf, ld, self._raw_grad(), self.ld, self.m, self.c, 1.0, 1.0 if self._written() else 0.0)
            self.gwritten = True

    def defer_residual(self, buf: H.DeviceBuffer, ld: int) -> bool:
        """Residual-Add backward: instead of copying the Add's output gradient into this (block input) gradient now, let the
        next writer -- the expand conv's backward-data GEMM -- add it in its epilogue.  False: not possible, copy as usual."""
        if self.parent is not None or self._written() or self.pending is not None or self.split_parent:
            return False
        self.pending = (buf, ld)
        return True

    def grad_slot(self) -> Tuple[H.DeviceBuffer, int]:
        """(gradient buffer, accumulate flag) for a consumer's backward; first writer overwrites."""
        self._materialise_pending()
        acc = 1 if self._written() else 0
        self.gwritten = True
        return self._raw_grad(), acc

    def grad_slot_residual(self):
        """as grad_slot for a writer whose kernel can add a residual tensor itself: (buffer, accumulate, residual, ldr)"""
        res, ldr = (None, 0)
        if self.pending is not None and not self._written():
            (res, ldr), self.pending = self.pending, None
        else:
            self._materialise_pending()
        acc = 1 if self._written() else 0
        self.gwritten = True
        return self._raw_grad(), acc, res, ldr


class BNRec:
    """Device state of one BatchNormalization layer."""

    def __init__(self, eng, layer: K.BatchNormalization, c: int, scale=None, shift=None):
        ctx = eng.ctx
        self.layer, self.c = layer, c
        self.gamma, self.beta = eng.param_view(layer, "gamma"), eng.param_view(layer, "beta")
        self.dgamma, self.dbeta = eng.grad_view(layer, "gamma"), eng.grad_view(layer, "beta")
        self.moving_mean, self.moving_var = eng.state_view(layer, "moving_mean"), eng.state_view(layer, "moving_variance")
        self.scale = scale if scale is not None else ctx.empty(c)
        self.shift = shift if shift is not None else ctx.empty(c)
        self.mean, self.invstd, self.k1, self.k0 = ctx.empty(c), ctx.empty(c), ctx.empty(c), ctx.empty(c)
        self.act = ACT_NONE
        self.bwd_done = False     # this step's backward reduction was already produced by a fused consumer kernel


class Val:
    """Lowered value of a symbolic tensor: a = act(scale * store + shift) (scale None -> act(store))."""

    def __init__(self, store: Store, scale=None, shift=None, act=ACT_NONE, bn: Optional[BNRec] = None, **meta):
        self.store, self.scale, self.shift, self.act, self.bn = store, scale, shift, act, bn
        self.meta = meta

    def view(self) -> H.ViewStruct:
        return H.view(self.store.buf, self.scale, self.shift, self.act)

    def gview(self) -> H.GViewStruct:
        """gradient view of this value's RAW store for the backward of the op that produced the store"""
        s = self.store
        if self.bn is not None:
            b = self.bn
            return H.gview(s.grad, s.buf, b.scale, b.shift, b.k1, b.k0, b.act)
        return H.gview(s.grad)


# ====================================================================================================== ops
class Op:
    name = ""
    reach = frozenset()      # names of the model outputs that depend on this op (set by Engine._emit)

    def fwd(self):
        pass

    def bwd(self):
        pass


class StemOp(Op):
    def __init__(self, eng, layer: K.Conv2D, image: Store, rescale, out: Store):
        self.e, self.layer, self.image, self.rescale, self.out = eng, layer, image, rescale, out
        self.name = layer.name
        self.w = eng.param_view(layer, "kernel")
        self.b = eng.param_view(layer, "bias") if layer.use_bias else None
        self.dw = eng.grad_view(layer, "kernel")
        self.db = eng.grad_view(layer, "bias") if layer.use_bias else None
        if eng.training and self.b is None:
            # (a biased stem -- ShuffleNetV2, models.py:619 -- gets its batch statistics from a ChannelStatsOp instead: the fused
            # epilogue relies on padded rows being exactly zero)
            out.nparts = eng.ctx.parts("ssdseg_stem_conv_parts", image.n, image.h, image.w, out.c)
            out.stats = eng.ctx.empty((out.nparts, 2, out.c))
        self.out_val: Optional[Val] = None

    def fwd(self):
        i = self.image
        self.e.ctx.call("ssdseg_stem_conv_fwd", i.buf, self.w, self.b, self.out.buf, i.n, i.h, i.w, i.c, self.out.c,
                        self.rescale[0], self.rescale[1], self.out.stats if self.b is None else None)

    def bwd(self):
        i = self.image
        db = self.db
        if db is not None and self.out_val.bn is not None:
            # a bias in front of a training-mode BatchNormalization has an identically zero gradient (the batch mean absorbs
            # it); the reference's autodiff produces round-off noise there
            db.zero_()
            db = None
        self.e.ctx.call("ssdseg_stem_conv_bwd_weight", i.buf, self.out_val.gview(), self.dw, db, i.n, i.h, i.w, i.c, self.out.c,
                        self.rescale[0], self.rescale[1])


class DwOp(Op):
    def __init__(self, eng, layer, wname: str, inp: Val, out: Store, stride: int, dilation: int):
        self.e, self.layer, self.inp, self.out, self.stride, self.dilation = eng, layer, inp, out, stride, dilation
        self.name = layer.name + ":dw"
        self.w, self.dw = eng.param_view(layer, wname), eng.grad_view(layer, wname)
        s = inp.store
        if eng.training:
            out.nparts = eng.ctx.parts("ssdseg_dwconv_parts", s.n, s.h, s.w, s.c, stride, dilation)
            out.stats = eng.ctx.empty((out.nparts, 2, out.c))
        assert s.ld == s.c and out.ld == out.c, "depthwise kernels work on dense (non-sliced) tensors"
        self.out_val: Optional[Val] = None

    def fwd(self):
        s = self.inp.store
        self.e.ctx.call("ssdseg_dwconv_fwd", self.inp.view(), self.w, self.out.buf, s.n, s.h, s.w, s.c, self.stride, self.dilation,
                        self.out.stats)

    # set by the lowering when this conv is the only consumer of a BatchNorm(+ReLU) output, or the FIRST of several in layer
    # order -- i.e. the last one to contribute to that gradient in the backward pass (a backbone tap that also feeds the heads)
    fuse_input_bn = False

    def bwd(self):
        s = self.inp.store
        dx, acc = (s.grad_slot() if s.need_grad else (None, 0))
        b = self.inp.bn
        if self.fuse_input_bn and b is not None and dx is not None:
            # the kernel that completes dx also reduces the producer BN's (dgamma, dbeta, k1, k0): no second pass over the wide tensor
            self.e.ctx.call("ssdseg_dwconv_bwd_bn", self.inp.view(), self.w, self.out_val.gview(), dx, self.dw, s.n, s.h, s.w, s.c,
                            self.stride, self.dilation, acc, b.mean, b.invstd, b.dgamma, b.dbeta, b.k1, b.k0)
            b.bwd_done = True
            return
        self.e.ctx.call("ssdseg_dwconv_bwd", self.inp.view(), self.w, self.out_val.gview(), dx, self.dw, s.n, s.h, s.w, s.c, self.stride,
                        self.dilation, acc)


class PwOp(Op):
    def __init__(self, eng, layer, wname: str, inp: Val, out: Store):
        self.e, self.layer, self.inp, self.out = eng, layer, inp, out
        self.name = layer.name + ":pw"
        self.w, self.dw = eng.param_view(layer, wname), eng.grad_view(layer, wname)
        self.m, self.k, self.n = inp.store.m, inp.store.c, out.c
        if eng.training:
            out.nparts = eng.ctx.parts("ssdseg_pwconv_parts", self.m, self.n)
            out.stats = eng.ctx.empty((out.nparts, 2, self.n))
        self.out_val: Optional[Val] = None
        # the forward tile GEMM streams Wt[n][k]: one batched transposition per step for all layers (Engine._refresh_wt)
        self.wt = eng.register_wt(self.w, self.m, inp.store.ld, self.k, self.n)

    fuse_input_bn = False   # set by the lowering when this conv is the only consumer of a (wide) BatchNorm(+ReLU) output

    def fwd(self):
        self.e.ctx.call("ssdseg_pwconv_fwd_wt", self.inp.view(), self.inp.store.ld, self.w, self.wt, self.out.buf, self.out.ld, self.m, self.k,
                        self.n, self.out.stats)

    def bwd(self):
        s = self.inp.store
        gv = self.out_val.gview()
        b = self.inp.bn
        if (self.fuse_input_bn and b is not None and s.need_grad and s.pending is None and not s._written()):
            # sole consumer of a BatchNorm(+ReLU6) output (the depthwise BN in front of a project conv): that BN's backward
            # reduction rides in this conv's backward-data epilogue instead of a separate pass over (g, y)
            dx, _ = s.grad_slot()
            self.e.ctx.call("ssdseg_pwconv_bwd_bn", self.inp.view(), s.ld, gv, self.out.ld, self.w, dx, s.ld, self.dw, self.m, self.k, self.n,
                            b.mean, b.invstd, b.dgamma, b.dbeta, b.k1, b.k0)
            b.bwd_done = True
            return
        if s.need_grad:
            # dx and dW in one call: for few input channels (the expand convs) one kernel reads the wide gradient once
            dx, acc, res, ldr = s.grad_slot_residual()
            self.e.ctx.call("ssdseg_pwconv_bwd", self.inp.view(), s.ld, gv, self.out.ld, self.w, dx, s.ld, self.dw, self.m, self.k, self.n,
                            res, ldr, acc)
        else:
            self.e.ctx.call("ssdseg_pwconv_bwd_weight", self.inp.view(), s.ld, gv, self.out.ld, self.dw, self.m, self.k, self.n)


class Conv3Op(Op):
    def __init__(self, eng, layer, inp: Val, out: Store):
        self.e, self.layer, self.inp, self.out = eng, layer, inp, out
        self.name = layer.name + ":conv3x3"
        self.w, self.dw = eng.param_view(layer, "kernel"), eng.grad_view(layer, "kernel")
        s = inp.store
        if eng.training:
            out.nparts = eng.ctx.parts("ssdseg_conv3x3_parts", s.n, s.h, s.w, s.c, out.c)
            out.stats = eng.ctx.empty((out.nparts, 2, out.c))
        assert out.ld == out.c
        self.out_val: Optional[Val] = None
        # large layers (Winograd kernels): the forward leaves act(BN(x)) in a zero-bordered copy which the weight gradient reads
        # again, so the view is applied once per step (ssdseg_conv3x3_fwd_saved / _bwd_weight_saved)
        self.xsaved = None
        self.fuse_input_bn = False   # set by the lowering (narrow convs that are the only consumer of a BatchNorm output)
        if eng.training:
            need = C.c_longlong()
            H._check(eng.ctx.lib.ssdseg_conv3x3_saved_floats(s.n, s.h, s.w, s.c, out.c, C.byref(need)), "ssdseg_conv3x3_saved_floats")
            if need.value > 0:
                self.xsaved = eng.ctx.zeros(need.value)       # (the border stays zero for the life of the buffer)
        # The input of the DeepLabV3+ decoder conv is Concatenate(up-sampled ASPP output, backbone branch) (blocks.py:104-113).  The
        # up-sampling writes final (already activated) values into channels [0, c) of the concat buffer, which the saved-input
        # forward then copies into the zero-bordered buffer.  Nothing else reads those channels of the plain concat buffer, so the
        # up-sampling writes them into the bordered buffer directly and the copy shrinks to the other branch's channels.
        self.saved_from = 0
        if self.xsaved is not None and s.ld == s.c:
            up = [op for op in eng.ops if isinstance(op, BilinearOp) and op.out.parent is s and op.out.coff == 0 and op.out.ld == s.c
                  and op.out.h == s.h and op.out.w == s.w]
            users = [op for op in eng.ops if any((getattr(v, "store", v) is s) for v in vars(op).values())]
            sole = len(eng.cons.get(id(layer.inbound[0]), [])) == 1 and id(layer.inbound[0]) not in {id(t) for t in eng.model.outputs}
            if len(up) == 1 and not users and sole:
                up[0].padded_out = (self.xsaved, s.c)
                self.saved_from = up[0].out.c

    def fwd(self):
        s = self.inp.store
        if self.xsaved is not None:
            c_from = self.saved_from if os.environ.get("SSDSEG_CONV3_PADFUSE", "1") != "0" else 0
            self.e.ctx.call("ssdseg_conv3x3_fwd_saved_from", self.inp.view(), s.ld, self.w, self.out.buf, s.n, s.h, s.w, s.c, self.out.c,
                            self.out.stats, self.xsaved, c_from)
            return
        self.e.ctx.call("ssdseg_conv3x3_fwd", self.inp.view(), s.ld, self.w, self.out.buf, s.n, s.h, s.w, s.c, self.out.c, self.out.stats)

    def bwd(self):
        s = self.inp.store
        gv = self.out_val.gview()
        if self.out_val.bn is not None:
            # both backward kernels read every dy element many times (9 taps): form dy = BN-backward(g, y) once, in place over g
            # (nothing else reads this g afterwards), and hand them the single-tensor identity view
            o = self.out
            self.e.ctx.call("ssdseg_gview_materialize", gv, o.ld, o.m, o.c)
            gv = H.gview(o.grad)
        # dW is off the critical path, but for a wide conv both backward kernels are bound by the SAME resource (the matrix pipes,
        # 0.8 of peak each on their own): side by side they only split the CUs and thrash each other's L2.  The side stream is
        # for pairing a weight gradient with an HBM-bound neighbour; here it is used only for the narrow (HBM-bound) 256 -> 4 conv.
        side = self.out.c <= 8 if os.environ.get("SSDSEG_CONV3_SIDE") is None else os.environ["SSDSEG_CONV3_SIDE"] == "1"
        if side:
            self.e.ctx.side(True)
        try:
            if self.xsaved is not None and self.out_val.bn is not None:     # (materialised dy: the saved-input weight gradient's form)
                self.e.ctx.call("ssdseg_conv3x3_bwd_weight_saved", self.xsaved, self.out.grad, self.dw, s.n, s.h, s.w, s.c, self.out.c)
            else:
                self.e.ctx.call("ssdseg_conv3x3_bwd_weight", self.inp.view(), s.ld, gv, self.dw, s.n, s.h, s.w, s.c, self.out.c)
        finally:
            if side:
                self.e.ctx.side(False)
        b = self.inp.bn
        if (self.fuse_input_bn and b is not None and s.need_grad and s.pending is None and not s._written()):
            # sole consumer of a BatchNorm(+ReLU) output (the decoder's logits conv behind sepconv-batchnorm): that BN's backward
            # sums ride in the epilogue of the GEMM that writes dx
            dx, _ = s.grad_slot()
            self.e.ctx.call("ssdseg_conv3x3_bwd_data_bn", self.inp.view(), gv, self.w, dx, s.ld, s.n, s.h, s.w, s.c, self.out.c,
                            b.mean, b.invstd, b.dgamma, b.dbeta, b.k1, b.k0)
            b.bwd_done = True
        elif s.need_grad:
            dx, acc = s.grad_slot()
            self.e.ctx.call("ssdseg_conv3x3_bwd_data", gv, self.w, dx, s.ld, s.n, s.h, s.w, s.c, self.out.c, acc)


class BnOp(Op):
    def __init__(self, eng, rec: BNRec, store: Store):
        self.e, self.rec, self.store = eng, rec, store
        self.name = rec.layer.name

    def fwd(self):
        r, s = self.rec, self.store
        l = r.layer
        if self.e.training:
            assert s.stats is not None, f"{self.name}: producer of {s.name} did not emit batch statistics"
            self.e.ctx.call("ssdseg_bn_finalize", s.stats, s.nparts, r.c, float(s.m), r.gamma, r.beta, l.epsilon, l.momentum, r.moving_mean,
                            r.moving_var, r.mean, r.invstd, r.scale, r.shift, 1)
        else:
            self.e.ctx.call("ssdseg_bn_finalize", None, 0, r.c, 0.0, r.gamma, r.beta, l.epsilon, l.momentum, r.moving_mean, r.moving_var,
                            None, None, r.scale, r.shift, 0)

    def bwd(self):
        r, s = self.rec, self.store
        if r.bwd_done:            # reduced by the consumer's backward kernel (DwOp.bwd)
            r.bwd_done = False
            return
        self.e.ctx.call("ssdseg_bn_bwd_reduce", s.grad, s.ld, s.buf, s.ld, s.m, r.c, r.scale, r.shift, r.mean, r.invstd, r.act, r.dgamma,
                        r.dbeta, r.k1, r.k0)


class ChannelStatsOp(Op):
    """batch statistics for a raw tensor whose producer has no fused stats epilogue"""

    def __init__(self, eng, store: Store):
        self.e, self.store = eng, store
        store.nparts = eng.ctx.parts("ssdseg_channel_stats_parts", store.m, store.c)
        store.stats = eng.ctx.empty((store.nparts, 2, store.c))

    def fwd(self):
        s = self.store
        self.e.ctx.call("ssdseg_channel_stats", s.buf, s.ld, s.m, s.c, s.stats)


class ApplyOp(Op):
    """out = view(a) (+ view(b)): Add, materialisation, copies into concat slices"""

    def __init__(self, eng, a: Val, b: Optional[Val], out: Store, name: str):
        self.e, self.a, self.b, self.out, self.name = eng, a, b, out, name

    def fwd(self):
        a, b, o = self.a, self.b, self.out
        self.e.ctx.call("ssdseg_bn_apply", a.view(), a.store.ld, b.view() if b is not None else None, b.store.ld if b is not None else 0,
                        o.buf, o.ld, o.m, o.c)

    alias_a = False   # Add: `a` (a projection's BN output consumed only here) shares this op's output-gradient buffer

    def bwd(self):
        o = self.out
        og = o.grad
        for v in (self.a, self.b):
            if v is None or not v.store.need_grad:
                continue
            # d(out)/d(activated value) = 1 (the activation mask is applied by the producer's BN/ReLU backward)
            if v is self.a and self.alias_a:
                v.store.gwritten = True            # dL/da IS dL/dout: same buffer, no copy
                continue
            if v is self.b and self.b is not None and v.store.defer_residual(og, o.ld):
                continue                           # added by the next writer's GEMM epilogue (or on first read)
            g, acc = v.store.grad_slot()
            self.e.ctx.call("ssdseg_axpby", og, o.ld, g, v.store.ld, o.m, o.c, 1.0, 1.0 if acc else 0.0)


class ActBwdOp(Op):
    """backward-only: g *= act'(x) in place for an activation that sits on a materialised tensor"""

    def __init__(self, eng, store: Store, act: int, name):
        self.e, self.store, self.act, self.name = eng, store, act, name

    def bwd(self):
        s = self.store
        self.e.ctx.call("ssdseg_act_bwd", s.grad, s.ld, s.buf, s.ld, s.m, s.c, self.act)


class MaxPoolOp(Op):
    """MaxPooling2D(3, strides=2, 'same') (reference models.py:629)"""

    def __init__(self, eng, inp: Val, out: Store, name):
        self.e, self.inp, self.out, self.name = eng, inp, out, name

    def fwd(self):
        s = self.inp.store
        self.e.ctx.call("ssdseg_maxpool3x3s2_fwd", self.inp.view(), self.out.buf, s.n, s.h, s.w, s.c)

    def bwd(self):
        s = self.inp.store
        if s.need_grad:
            dx, acc = s.grad_slot()
            assert acc == 0, "max-pool input with another consumer"
            self.e.ctx.call("ssdseg_maxpool3x3s2_bwd", self.inp.view(), self.out.grad, dx, s.n, s.h, s.w, s.c)


class ShuffleOp(Op):
    """channel shuffle: Reshape(h, w, g, c/g) -> Permute(1, 2, 4, 3) -> Reshape(h, w, c) (reference models.py:497-503)"""

    def __init__(self, eng, inp: Val, out: Store, groups: int, name):
        self.e, self.inp, self.out, self.groups, self.name = eng, inp, out, groups, name

    def fwd(self):
        i, o = self.inp.store, self.out
        self.e.ctx.call("ssdseg_channel_shuffle", self.inp.view(), i.ld, o.buf, o.ld, i.m, i.c, self.groups, 0)

    def bwd(self):
        i, o = self.inp.store, self.out
        g, acc = i.grad_slot()     # gradient w.r.t. the ACTIVATED concat values; the branches' BN backward applies the masks
        assert acc == 0
        self.e.ctx.call("ssdseg_channel_shuffle", H.view(o.grad), o.ld, g, i.ld, i.m, i.c, self.groups, 1)


def pad4(c: int) -> int:
    """physical channel count of a tensor with c logical channels: the kernels work on 16-byte channel vectors, so a branch
    of 58 (ShuffleNetV2 '1x') or 122 ('2x') channels lives in a zero-padded 60- / 124-channel tensor"""
    return (int(c) + 3) // 4 * 4


class SplitGatherOp(Op):
    """channel Split whose parts are not multiples of 4 channels wide (ShuffleNetV2 '1x' / '2x', reference models.py:573):
    each part is gathered into its own zero-padded dense tensor; backward gathers the parts' gradients back"""

    def __init__(self, eng, inp: Val, outs: List[Store], widths: List[int], name):
        self.e, self.inp, self.outs, self.name = eng, inp, outs, name
        cin = inp.store.c
        self.tf, self.tb = [], []
        off = 0
        for st, w in zip(outs, widths):
            self.tf.append(eng.ctx.array(np.array([off + i if i < w else -1 for i in range(st.c)], np.int32)))
            self.tb.append(eng.ctx.array(np.array([j - off if off <= j < off + w else -1 for j in range(cin)], np.int32)))
            off += w

    def fwd(self):
        i = self.inp.store
        for st, t in zip(self.outs, self.tf):
            self.e.ctx.call("ssdseg_channel_gather", self.inp.view(), i.ld, st.buf, st.ld, i.m, st.c, t, 0)

    def bwd(self):
        i = self.inp.store
        g, acc = i.grad_slot()
        for k, (st, t) in enumerate(zip(self.outs, self.tb)):
            self.e.ctx.call("ssdseg_channel_gather", H.view(st.grad), st.ld, g, i.ld, i.m, i.c, t, 1 if (acc or k > 0) else 0)


class TableShuffleOp(Op):
    """channel shuffle of a concat whose parts carry padding channels: the permutation (reference models.py:497-503) composed
    with padded -> packed re-indexing, as one table lookup (the lazily fused BN + ReLU of the branches rides along, as in
    ShuffleOp); backward scatters through the inverse table (padding channels get zero gradient)"""

    def __init__(self, eng, inp: Val, out: Store, groups: int, name):
        self.e, self.inp, self.out, self.name = eng, inp, out, name
        parts = inp.store.logical_parts
        phys_of = [off + i for off, cl in parts for i in range(cl)]
        c_log = len(phys_of)
        assert c_log % groups == 0 and pad4(c_log) == out.c
        fwd = [phys_of[(j % groups) * (c_log // groups) + j // groups] if j < c_log else -1 for j in range(out.c)]
        inv = [-1] * inp.store.c
        for j, p in enumerate(fwd):
            if p >= 0:
                inv[p] = j
        self.tf, self.tb = eng.ctx.array(np.array(fwd, np.int32)), eng.ctx.array(np.array(inv, np.int32))

    def fwd(self):
        i, o = self.inp.store, self.out
        self.e.ctx.call("ssdseg_channel_gather", self.inp.view(), i.ld, o.buf, o.ld, i.m, o.c, self.tf, 0)

    def bwd(self):
        i, o = self.inp.store, self.out
        g, acc = i.grad_slot()
        assert acc == 0
        self.e.ctx.call("ssdseg_channel_gather", H.view(o.grad), o.ld, g, i.ld, i.m, i.c, self.tb, 0)


class GapOp(Op):
    def __init__(self, eng, inp: Val, out: Store, name):
        self.e, self.inp, self.out, self.name = eng, inp, out, name
        assert inp.store.ld == inp.store.c

    def fwd(self):
        s = self.inp.store
        self.e.ctx.call("ssdseg_gap_fwd", self.inp.view(), self.out.buf, s.n, s.h * s.w, s.c)

    def bwd(self):
        s = self.inp.store
        if s.need_grad:
            dx, acc = s.grad_slot()
            self.e.ctx.call("ssdseg_gap_bwd", self.out.grad, dx, s.n, s.h * s.w, s.c, acc)


class BilinearOp(Op):
    def __init__(self, eng, inp: Val, out: Store, fy: int, fx: int, name):
        self.e, self.inp, self.out, self.fy, self.fx, self.name = eng, inp, out, fy, fx, name

    # set by a Conv3Op that keeps a zero-bordered copy of its (concatenated) input: (buffer, row stride in floats) -- this op's
    # output then goes straight into the interior of that copy and the plain concat slice is not written (Conv3Op.__init__)
    padded_out = None

    def fwd(self):
        s = self.inp.store
        if self.padded_out is not None and os.environ.get("SSDSEG_CONV3_PADFUSE", "1") != "0":
            buf, ld = self.padded_out
            self.e.ctx.call("ssdseg_bilinear_fwd_padded", self.inp.view(), s.ld, buf, ld, s.n, s.h, s.w, s.c, self.fy, self.fx)
            return
        self.e.ctx.call("ssdseg_bilinear_fwd", self.inp.view(), s.ld, self.out.buf, self.out.ld, s.n, s.h, s.w, s.c, self.fy, self.fx)

    def bwd(self):
        s = self.inp.store
        if s.need_grad:
            dx, acc = s.grad_slot()
            self.e.ctx.call("ssdseg_bilinear_bwd", self.out.grad, self.out.ld, dx, s.ld, s.n, s.h, s.w, s.c, self.fy, self.fx, acc)


class MaskHeadOp(Op):
    """logits --bilinear x(fy,fx)--> softmax (= output-mask) [--> weighted cross-entropy]"""

    def __init__(self, eng, logits: Val, fy, fx, name, prob: Optional[Store]):
        self.e, self.logits, self.fy, self.fx, self.name, self.prob = eng, logits, fy, fx, name, prob
        s = logits.store
        assert logits.scale is None and logits.act == ACT_NONE and s.ld == s.c
        self.y_true: Optional[H.DeviceBuffer] = None
        self.class_weights: Optional[H.DeviceBuffer] = None
        self.loss: Optional[H.DeviceBuffer] = None
        self.loss_scale = 0.0
        self.kind = "cross_entropy"                       # | "dice" | "dice_square" (reference losses.py:175-307)
        self.coef: Optional[H.DeviceBuffer] = None        # dice: per-image backward coefficients left by the forward

    def fwd(self):
        s = self.logits.store
        prob = self.prob.buf if self.prob is not None else None
        if self.loss is not None and self.kind != "cross_entropy":
            self.e.ctx.call("ssdseg_mask_head_fwd_dice", s.buf, s.n, s.h, s.w, s.c, self.fy, self.fx, self.y_true, self.class_weights,
                            1 if self.kind == "dice_square" else 0, prob, self.loss, self.coef)
            return
        self.e.ctx.call("ssdseg_mask_head_fwd", s.buf, s.n, s.h, s.w, s.c, self.fy, self.fx, self.y_true if self.loss is not None else None,
                        self.class_weights, prob, self.loss)

    def bwd(self):
        if self.loss is None:
            return
        s = self.logits.store
        g, acc = s.grad_slot()
        assert acc == 0
        if self.kind != "cross_entropy":
            self.e.ctx.call("ssdseg_mask_head_bwd_dice", s.buf, s.n, s.h, s.w, s.c, self.fy, self.fx, self.y_true, self.coef,
                            1 if self.kind == "dice_square" else 0, self.loss_scale, g)
            return
        self.e.ctx.call("ssdseg_mask_head_bwd", s.buf, s.n, s.h, s.w, s.c, self.fy, self.fx, self.y_true, self.class_weights, self.loss_scale, g)


class HeadGatherOp(Op):
    """act(bn(head)) of one SSD feature map, viewed as (B, H*W*boxes, 4), written at its anchor offset of the
    (B, anchors, 4) concat (reference blocks.py:155 Reshape + models.py:256/271 Concatenate axis=1)"""

    def __init__(self, eng, inp: Val, out: Store, anchor_offset: int, anchors_total: int, name):
        self.e, self.inp, self.out, self.off, self.total, self.name = eng, inp, out, anchor_offset, anchors_total, name

    def fwd(self):
        s = self.inp.store
        self.e.ctx.call("ssdseg_head_gather", self.inp.view(), self.out.buf, s.n, s.h * s.w * s.c, s.c, self.off * 4, self.total * 4, 0)

    def bwd(self):
        s = self.inp.store
        g, acc = s.grad_slot()
        assert acc == 0
        self.e.ctx.call("ssdseg_head_gather", H.view(self.out.grad), g, s.n, s.h * s.w * s.c, s.c, self.off * 4, self.total * 4, 1)


class SoftmaxRowsOp(Op):
    def __init__(self, eng, inp: Store, out: Store, name):
        self.e, self.inp, self.out, self.name = eng, inp, out, name

    def fwd(self):
        self.e.ctx.call("ssdseg_softmax_rows", H.view(self.inp.buf), self.out.buf, self.inp.m, self.inp.c)


class DetLossOp(Op):
    """confidence + localization losses; the gradients w.r.t. the pre-softmax logits / box offsets are produced in
    the forward call (fused) and simply stay in the concat stores' gradient buffers."""

    def __init__(self, eng, logits: Store, probs: Store, boxes: Store):
        self.e, self.logits, self.probs, self.boxes = eng, logits, probs, boxes
        self.name = "det-loss"
        b = probs.n
        self.y_labels = self.y_boxes = None
        self.conf_loss, self.loc_loss = eng.ctx.empty(b), eng.ctx.empty(b)
        self.w_conf = self.w_loc = 1.0

    def fwd(self):
        p = self.probs
        assert self.w_conf == self.w_loc, "per-output loss weights must match for the fused detection loss"
        tr = self.e.training
        self.e.ctx.call("ssdseg_det_loss", self.y_labels, p.buf, self.y_boxes, self.boxes.buf, p.n, p.h * p.w, p.c, self.w_conf / p.n,
                        self.conf_loss, self.loc_loss, self.logits.grad if tr else None, self.boxes.grad if tr else None, None)

    def bwd(self):
        self.logits.gwritten = True
        self.boxes.gwritten = True


class DecodeNmsOp(Op):
    def __init__(self, eng, boxes: Store, probs: Store, decode: L.DecodeBoxesCentroidsOffsets, nms: L.NonMaximumSuppression,
                 suppress_with: Optional[Store], out: Store):
        self.e, self.boxes, self.probs, self.nms, self.mask, self.out = eng, boxes, probs, nms, suppress_with, out
        ctx = eng.ctx
        a = boxes.h * boxes.w
        cent = np.stack([decode.center_x_boxes_default, decode.center_y_boxes_default, decode.width_boxes_default,
                         decode.height_boxes_default], axis=1).astype(np.float32)
        assert cent.shape == (a, 4)
        self.centroids = ctx.array(cent)
        self.stds = (C.c_float * 4)(decode.standard_deviation_center_x_offsets, decode.standard_deviation_center_y_offsets,
                                    decode.standard_deviation_width_offsets, decode.standard_deviation_height_offsets)
        self.corners = ctx.empty((boxes.n, a, 4))
        self.probs_s = ctx.empty((probs.n, a, probs.c)) if suppress_with is not None else None
        self.valid = ctx.empty(boxes.n, np.int32)
        self.name = nms.name

    def fwd(self):
        b, a, c = self.boxes.n, self.boxes.h * self.boxes.w, self.probs.c
        ctx = self.e.ctx
        ctx.call("ssdseg_decode_boxes", self.boxes.buf, self.centroids, b, a, self.stds, self.corners)
        probs = self.probs.buf
        if self.mask is not None:
            m = self.mask
            ctx.call("ssdseg_seg_suppress", m.buf, m.m, m.c, probs, b * a, self.probs_s)
            probs = self.probs_s
        n = self.nms
        ctx.call("ssdseg_combined_nms", self.corners, probs, b, a, c, n.max_number_of_boxes_per_class, n.max_number_of_boxes_per_sample,
                 float(n.boxes_iou_threshold), float(n.labels_probability_threshold), self.out.buf, self.valid)


# ====================================================================================================== engine
class Engine:
    """One lowered instance of a model for a fixed batch size and mode."""

    def __init__(self, model: K.Model, batch_size: int, training: bool, ctx: Optional[H.Context] = None, grad_bucket=None):
        self.model, self.batch, self.training = model, int(batch_size), bool(training)
        self.ctx = ctx if ctx is not None else default_context()
        self.ops: List[Op] = []
        self.vals: Dict[int, Val] = {}          # id(KTensor) -> Val
        self.stores: List[Store] = []
        self.output_vals: List[Val] = []
        self.loss_ops: Dict[str, Op] = {}
        self._wt_rows: List[Tuple[int, int, int, int]] = []
        self._wt_table = None
        self._alloc_params(grad_bucket)
        self._plan_concats()
        reach = self._outputs_reached()
        self._cur_reach = frozenset()
        for layer in model.layers:
            self._cur_reach = reach[id(layer)]
            self._lower(layer)
        self.output_vals = [self.vals[id(t)] for t in model.outputs]
        self.adam_step_count = 0

    # ------------------------------------------------------------------ parameters
    def _alloc_params(self, grad_bucket):
        holder = getattr(self.model, "_trained", None) or self.model   # an inference model shares its trained model's weights
        owner = getattr(holder, "_engine_params", None)
        if owner is None:
            tr, st = [], []
            for l in self.model.layers:
                for wname, arr in l.weights.items():
                    (tr if (wname in l.trainable_names) else st).append((l, wname, arr))
            n_tr = sum(a.size for _, _, a in tr)
            n_st = sum(a.size for _, _, a in st)
            ctx = self.ctx
            owner = dict(index={}, n_tr=n_tr, n_st=n_st)
            flat = np.concatenate([a.reshape(-1) for _, _, a in tr]) if tr else np.zeros(0, np.float32)
            flat_s = np.concatenate([a.reshape(-1) for _, _, a in st]) if st else np.zeros(0, np.float32)
            owner["params"] = ctx.array(flat.astype(np.float32))
            owner["state"] = ctx.array(flat_s.astype(np.float32))
            off = 0
            for l, wname, arr in tr:
                owner["index"][(id(l), wname)] = ("params", off, arr.shape)
                off += arr.size
            off = 0
            for l, wname, arr in st:
                owner["index"][(id(l), wname)] = ("state", off, arr.shape)
                off += arr.size
            owner["grads"] = owner["adam_m"] = owner["adam_v"] = None
            holder._engine_params = owner
            for l in self.model.layers:
                if l.weights:
                    l._engine_sync = (self._pull_layer, self._push_layer)
        self.P = owner
        if self.training and owner["grads"] is None:
            n = owner["n_tr"]
            owner["grads"] = grad_bucket if grad_bucket is not None else self.ctx.zeros(n)
            owner["adam_m"], owner["adam_v"] = self.ctx.zeros(n), self.ctx.zeros(n)

    def _view(self, bucket: str, layer, wname) -> H.DeviceBuffer:
        which, off, shape = self.P["index"][(id(layer), wname)]
        src = {"params": self.P[which], "grads": self.P["grads"] if which == "params" else None}[bucket]
        if src is None:
            return None
        return src.view(off, shape)

    @staticmethod
    def _phys_shape(wname: str, shape) -> Tuple[int, ...]:
        """shape of a weight as the kernels see it: channel dimensions rounded up to multiples of 4 (only ShuffleNetV2 '1x' /
        '2x' have any that are not: 58 / 122); taps, the 3 image channels and depth multipliers stay"""
        s = tuple(int(x) for x in shape)
        if len(s) == 1:                                     # BatchNormalization vectors, biases
            return (pad4(s[0]),)
        if wname == "depthwise_kernel":                     # (kh, kw, c, 1)
            return (s[0], s[1], pad4(s[2]), s[3])
        if len(s) == 4 and s[0] == 1 and s[1] == 1:         # pointwise kernels (1, 1, cin, cout)
            return (1, 1, pad4(s[2]), pad4(s[3]))
        return s

    def _bucket_view(self, layer, wname, grads=False):
        which, off, shape = self.P["index"][(id(layer), wname)]
        return (self.P["grads"] if grads else self.P[which]).view(off, shape)

    def _padded_view(self, layer, wname, kind: str):
        """zero-padded device copy of a weight (kind "p": parameter / state, refreshed from the bucket before every forward pass;
        "g": gradient, folded back into the bucket after the backward pass)"""
        which, off, shape = self.P["index"][(id(layer), wname)]
        phys = self._phys_shape(wname, shape)
        reg = self.__dict__.setdefault("_padded", {})
        key = (id(layer), wname, kind)
        if key not in reg:
            dims = [d for d in shape if d != 1] or [1]
            pdims = [d for d in phys if d != 1] or [1]
            rows = int(np.prod(dims[:-1])) if len(dims) > 1 else 1
            reg[key] = dict(buf=self.ctx.zeros(phys), rows=rows, cols=dims[-1], lds=dims[-1], ldd=pdims[-1], layer=layer, wname=wname,
                            which=which)
        return reg[key]["buf"]

    def register_wt(self, w: H.DeviceBuffer, m: int, ldx: int, k: int, n: int):
        """device buffer for the transposed copy Wt[n][k] of a pointwise kernel, refreshed by `_refresh_wt` at the start of every
        forward pass (None where the library's default dispatch for this shape does not stream a transposed copy)"""
        if self.ctx.parts("ssdseg_pwconv_wt_floats", m, ldx, k, n) == 0:
            return None
        wt = self.ctx.empty((n, k))
        self._wt_rows.append((w.ptr, wt.ptr, k, n))
        self._wt_table = None
        return wt

    def _refresh_wt(self):
        if not self._wt_rows:
            return
        if self._wt_table is None:
            self._wt_table = self.ctx.array(np.asarray(self._wt_rows, dtype=np.int64))
        tiles = max(-(-k // 32) * -(-n // 32) for _, _, k, n in self._wt_rows)
        self.ctx.call("ssdseg_transpose_batch", self._wt_table, len(self._wt_rows), tiles, sum(k * n for _, _, k, n in self._wt_rows))

    def param_view(self, layer, wname):
        which, off, shape = self.P["index"][(id(layer), wname)]
        if self._phys_shape(wname, shape) != tuple(shape):
            return self._padded_view(layer, wname, "p")
        return self.P[which].view(off, shape)

    state_view = param_view

    def grad_view(self, layer, wname):
        if not self.training:
            return None
        which, off, shape = self.P["index"][(id(layer), wname)]
        assert which == "params"
        if self._phys_shape(wname, shape) != tuple(shape):
            return self._padded_view(layer, wname, "g")
        return self.P["grads"].view(off, shape)

    def grad_array(self, layer, wname) -> np.ndarray:
        """this step's gradient of a weight in its Keras shape (tests)"""
        return self._bucket_view(layer, wname, grads=True).download()

    def _sync_padded(self, kind: str, to_bucket: bool, which: Optional[str] = None):
        """zero-padded copies of odd-width weights <-> their Keras-shaped slots of the flat bucket: ONE table-driven launch per call
        (ssdseg_copy2d_batch; the table is built once per (kind, direction, which) -- the pointers never change).  Was one
        ssdseg_copy2d per tensor: 160 launches per ShuffleNetV2-1x step (SSDSEG_COPY2D_BATCH=0 keeps that, bit-identical)."""
        padded = self.__dict__.get("_padded", {})
        if not padded:
            return
        key = (kind, to_bucket, which, len(padded))            # (a table made before the last padded tensor was registered is not reused)
        cache = self.__dict__.setdefault("_padded_tables", {})
        if key not in cache:
            rows = []
            for (lid, wname, k), r in padded.items():
                if k != kind or (which is not None and r["which"] != which):
                    continue
                bv = self._bucket_view(r["layer"], wname, grads=(kind == "g"))
                if to_bucket:
                    rows.append((bv.ptr, r["lds"], r["buf"].ptr, r["ldd"], r["rows"], r["cols"]))
                else:
                    rows.append((r["buf"].ptr, r["ldd"], bv.ptr, r["lds"], r["rows"], r["cols"]))
            cache[key] = (rows, self.ctx.array(np.asarray(rows, dtype=np.int64)) if rows else None)
        rows, table = cache[key]
        if not rows:
            return
        if os.environ.get("SSDSEG_COPY2D_BATCH", "1") == "0":
            for dst, ldd, src, lds, nr, nc in rows:
                self.ctx.call("ssdseg_copy2d", C.c_void_p(dst), ldd, C.c_void_p(src), lds, nr, nc)
            return
        self.ctx.call("ssdseg_copy2d_batch", table, len(rows), max(nr * nc for *_, nr, nc in rows), sum(nr * nc for *_, nr, nc in rows))

    def _pull_layer(self, layer):
        self.ctx.sync()
        for wname in layer.weights:
            layer.weights[wname] = self._bucket_view(layer, wname).download()

    def _push_layer(self, layer):
        for wname, arr in layer.weights.items():
            self._bucket_view(layer, wname).upload(arr)

    # ------------------------------------------------------------------ planning helpers
    def _consumers(self) -> Dict[int, List[K.Layer]]:
        cons: Dict[int, List[K.Layer]] = {}
        for l in self.model.layers:
            for t in l.inbound:
                cons.setdefault(id(t), []).append(l)
        return cons

    def _plan_concats(self):
        """channel-axis Concatenate: let each input's storage producer write straight into its slice"""
        self.cons = self._consumers()
        self.placement: Dict[int, Tuple[K.Concatenate, int]] = {}   # id(producer layer) -> (concat layer, channel offset)
        self.concat_store: Dict[int, Store] = {}
        outs = {id(t) for t in self.model.outputs}
        for l in self.model.layers:
            if not isinstance(l, K.Concatenate) or l.axis_resolved != 3:
                continue
            off = 0
            for t in l.inbound:
                c = pad4(t.shape[-1])
                cur, ok = t, True
                # walk back through lazily-fused per-channel layers to whoever writes memory
                while isinstance(cur.layer, (K.ReLU, K.BatchNormalization)):
                    if len(self.cons.get(id(cur), [])) != 1 or id(cur) in outs:
                        ok = False
                        break
                    cur = cur.layer.inbound[0]
                prod = cur.layer
                if ok and isinstance(prod, (K.Conv2D, K.SeparableConv2D, K.UpSampling2D)) and len(self.cons.get(id(cur), [])) == 1 \
                        and id(cur) not in outs and not (isinstance(prod, K.Conv2D) and prod.kernel_size == (3, 3)):
                    self.placement[id(prod)] = (l, off)
                off += c

    def _concat_parts(self, concat: K.Concatenate):
        """(store, wide scale, wide shift) of a channel concat, created on first use"""
        key = id(concat)
        if key not in self.concat_store:
            n, h, w, _ = (self.batch,) + tuple(concat.outputs[0].shape[1:])
            parts, c = [], 0                       # physical layout: every input at its padded width
            for t in concat.inbound:
                parts.append((c, int(t.shape[-1])))
                c += pad4(t.shape[-1])
            st = Store(self, n, h, w, c, concat.name)
            st.logical_parts = parts
            st.wide_scale = self.ctx.array(np.ones(c, np.float32))
            st.wide_shift = self.ctx.array(np.zeros(c, np.float32))
            self.concat_store[key] = st
            self.stores.append(st)
        return self.concat_store[key]

    def _out_store(self, layer, shape, name=None) -> Tuple[Store, Optional[H.DeviceBuffer], Optional[H.DeviceBuffer]]:
        """fresh dense store for a layer's raw output, or its slice of a concat buffer"""
        n, h, w, c = (self.batch,) + tuple(shape[1:])
        c = pad4(c)                                # (== c for every model but ShuffleNetV2 '1x' / '2x')
        if id(layer) in self.placement:
            concat, off = self.placement[id(layer)]
            parent = self._concat_parts(concat)
            st = parent.slice(off, c, name or layer.name)
            st.slice_scale = parent.wide_scale.view(off, (c,))
            st.slice_shift = parent.wide_shift.view(off, (c,))
            return st
        st = Store(self, n, h, w, c, name or layer.name)
        self.stores.append(st)
        return st

    def _emit(self, op: Op):
        op.reach = self._cur_reach          # names of the model outputs that depend on this op (branch scheduling, _schedule)
        self.ops.append(op)
        self._plans = None
        return op

    # ------------------------------------------------------------------ two independent branches on two streams
    DET_OUTPUTS = frozenset(("output-labels", "output-boxes"))      # the reference's output names (models.py:259,271)

    def _outputs_reached(self) -> Dict[int, frozenset]:
        """id(layer) -> names of the model outputs that depend on it"""
        names = {id(t): t.layer.name for t in self.model.outputs}
        reach: Dict[int, frozenset] = {}
        for layer in reversed(self.model.layers):
            r = set()
            for t in layer.outputs:
                if id(t) in names:
                    r.add(names[id(t)])
                for c in self.cons.get(id(t), []):
                    r |= reach.get(id(c), frozenset())
            reach[id(layer)] = frozenset(r)
        return reach

    def _schedule(self):
        """The multi-task graph is a shared trunk with two independent branches.  Which is which follows from reachability, not
        from layer names: the mask output depends on the backbone only up to the block-13 expansion (ASPP tap) -- everything
        behind it (block 13's depthwise conv ... block 16, the extra feature maps, the eight SSDLite heads, gather, softmax,
        the detection losses: ~150 launches of 5-60 us on 30x40 ... 1x1 maps that leave most CUs idle) only feeds the detection
        outputs.  A training engine issues that DETECTION branch on the context's side stream -- forward right behind the
        trunk, backward right at the start -- so it runs beside the MASK branch (ASPP, decoder, mask head: few long kernels
        that fill the chip) instead of before / after it: 31.8 -> 30.5 ms per batch-32 step.  The backward issue order stays
        detection, mask, trunk = the reverse layer order for every tensor more than one of them touches (checked below), so every
        first-writer / accumulate decision and every summation order is the same and results are bit-identical with
        SSDSEG_DET_SIDE=0 (one stream, layer order).  Cross-stream hazards: a mask-branch backward op that accumulates into a
        trunk tensor's gradient the detection branch wrote first (the block-13 tap feeds the ASPP and the first SSD head) waits
        for the side stream (`join_before`); the trunk's backward joins anyway.
        -> (trunk, detection, mask, join_before)"""
        if not self.training or os.environ.get("SSDSEG_DET_SIDE", "1") == "0":      # (read per pass: A/B runs and tests flip it)
            return self.ops, [], [], set()
        if getattr(self, "_plans", None) is not None:
            return self._plans
        def touched(op):
            out = set()
            for v in vars(op).values():
                for x in (v if isinstance(v, (list, tuple)) else (v,)):
                    st = x.store if isinstance(x, Val) else (x if isinstance(x, Store) else None)
                    while st is not None:
                        out.add(id(st))
                        st = st.parent
            return out

        pos = {id(op): i for i, op in enumerate(self.ops)}
        touch = {id(op): touched(op) for op in self.ops}
        det = [op for op in self.ops if op.reach and op.reach <= self.DET_OUTPUTS]
        mask = [op for op in self.ops if op.reach and not (op.reach & self.DET_OUTPUTS)]
        serial = (list(self.ops), [], [], set())
        if not det or not mask:
            self._plans = serial
            return self._plans
        # A detection-branch op that reads a tensor the mask branch also reads, and comes BEFORE those readers in layer order,
        # is the LAST to contribute to that tensor's gradient in the backward pass -- and may rely on it (the depthwise conv of
        # block 13 completes the block-13 tap's gradient and takes its BatchNorm's backward sums over the completed tensor,
        # DwOp.fuse_input_bn).  Such ops stay with the trunk: issued before the fork in the forward pass, after the join in the
        # backward pass, at their place in layer order.
        mask_last = {}
        for op in mask:
            for sid in touch[id(op)]:
                mask_last[sid] = max(mask_last.get(sid, -1), pos[id(op)])
        keep = [op for op in det if any(pos[id(op)] < mask_last.get(sid, -1) for sid in touch[id(op)])]
        det = [op for op in det if op not in keep]
        det_ids, mask_ids = {id(o) for o in det}, {id(o) for o in mask}
        trunk = [op for op in self.ops if id(op) not in det_ids and id(op) not in mask_ids]
        # Validity of the three-phase order (else: one stream, layer order).  (1) A demoted op runs before the fork, so nothing it
        # reads may come from the detection branch: no detection op ahead of it in layer order touches a tensor it touches.  (2) The backward
        # pass writes a tensor's gradient in the order detection, mask, trunk; that is the reverse layer order -- same first
        # writer, same accumulation order, bit-identical sums -- iff in layer order every trunk op on that tensor precedes every
        # branch op on it and every mask op precedes every detection op.
        det_out = set().union(*[touch[id(op)] for op in det]) if det else set()
        ok = bool(det) and not any(touch[id(k)] & touch[id(d)] for k in keep for d in det if pos[id(d)] < pos[id(k)])
        first = {}
        for kind, ops_ in (("det", det), ("mask", mask)):
            for op in ops_:
                for sid in touch[id(op)]:
                    f = first.setdefault(sid, {})
                    f[kind + "_min"] = min(f.get(kind + "_min", 1 << 30), pos[id(op)])
                    f[kind + "_max"] = max(f.get(kind + "_max", -1), pos[id(op)])
        for op in trunk:
            for sid in touch[id(op)] & set(first):
                f = first[sid]
                if pos[id(op)] > min(f.get("det_min", 1 << 30), f.get("mask_min", 1 << 30)):
                    ok = False
        for f in first.values():
            if "det_min" in f and "mask_max" in f and f["mask_max"] > f["det_min"]:
                ok = False
        if not ok:
            self._plans = serial
            return self._plans
        det_touch = det_out
        join_before = {id(op) for op in mask if touch[id(op)] & det_touch}
        self._plans = (trunk, det, mask, join_before)
        return self._plans

    def _dense(self, v: Val, name: str) -> Val:
        """a Val over a dense store (materialise sliced / strided inputs for kernels that need ld == c)"""
        if v.store.ld == v.store.c:
            return v
        s = v.store
        st = Store(self, s.n, s.h, s.w, s.c, name + ":dense")
        self.stores.append(st)
        self._emit(ApplyOp(self, v, None, st, name + ":dense"))
        return Val(st)

    # ------------------------------------------------------------------ lowering
    def _lower(self, layer: K.Layer):
        ins = [self.vals[id(t)] for t in layer.inbound]
        out_t = layer.outputs[0] if layer.outputs else None
        setv = lambda v: self.vals.__setitem__(id(out_t), v)

        if isinstance(layer, K.InputLayer):
            n, h, w, c = (self.batch,) + tuple(out_t.shape[1:])
            st = Store(self, n, h, w, c, layer.name, need_grad=False)
            self.stores.append(st)
            self.input_store = st
            setv(Val(st))
        elif isinstance(layer, K.Rescaling):
            setv(Val(ins[0].store, rescale=(layer.scale, layer.offset)))
        elif isinstance(layer, K.Conv2D):
            self._lower_conv(layer, ins[0], setv)
        elif isinstance(layer, K.DepthwiseConv2D):
            assert layer.strides[0] == layer.strides[1] and layer.dilation_rate[0] == layer.dilation_rate[1]
            st = self._out_store(layer, out_t.shape)
            op = self._emit(DwOp(self, layer, "depthwise_kernel", self._dense(ins[0], layer.name), st, layer.strides[0], layer.dilation_rate[0]))
            src = layer.inbound[0]
            cons = self.cons.get(id(src), [])
            op.fuse_input_bn = (op.inp is ins[0] and ins[0].bn is not None and ins[0].store.parent is None and bool(cons) and cons[0] is layer
                                and layer.dilation_rate[0] == 1)
            op.out_val = Val(st)
            setv(op.out_val)
            st.producer = op
        elif isinstance(layer, K.SeparableConv2D):
            assert layer.strides[0] == layer.strides[1] and layer.dilation_rate[0] == layer.dilation_rate[1]
            inp = self._dense(ins[0], layer.name)
            s = inp.store
            n, h, w, _ = (self.batch,) + tuple(out_t.shape[1:])
            mid = Store(self, n, h, w, s.c, layer.name + ":dw")
            self.stores.append(mid)
            dw = self._emit(DwOp(self, layer, "depthwise_kernel", inp, mid, layer.strides[0], layer.dilation_rate[0]))
            src = layer.inbound[0]
            # sole consumer of a BatchNorm(+ReLU) output (the decoder's sepconv behind the 3x3 conv): that BN's backward sums
            # ride in the depthwise backward, as for the MBConv blocks
            dw.fuse_input_bn = (inp is ins[0] and ins[0].bn is not None and ins[0].store.parent is None
                                and len(self.cons.get(id(src), [])) == 1 and id(src) not in {id(t) for t in self.model.outputs})
            mid.stats = None   # no BatchNormalization between the two halves
            dw.out_val = Val(mid)
            st = self._out_store(layer, out_t.shape)
            pw = self._emit(PwOp(self, layer, "pointwise_kernel", dw.out_val, st))
            pw.out_val = Val(st)
            st.producer = pw
            setv(pw.out_val)
        elif isinstance(layer, K.BatchNormalization):
            v = ins[0]
            assert v.scale is None and v.act == ACT_NONE and v.bn is None, f"{layer.name}: BatchNormalization input must be a raw conv output"
            st = v.store
            sc = getattr(st, "slice_scale", None)
            sh = getattr(st, "slice_shift", None)
            rec = BNRec(self, layer, st.c, sc, sh)
            if self.training and st.stats is None:
                self._emit(ChannelStatsOp(self, st))
            self._emit(BnOp(self, rec, st))
            nv = Val(st, rec.scale, rec.shift, ACT_NONE, rec)
            prod = getattr(st, "producer", None)
            if prod is not None:
                prod.out_val = nv
            setv(nv)
        elif isinstance(layer, K.ReLU):
            v = ins[0]
            act = _act_of(layer)
            assert v.act == ACT_NONE, f"{layer.name}: stacked activations are not used by ssdseglib"
            if v.bn is not None:
                assert len(self.cons.get(id(layer.inbound[0]), [])) == 1, f"{layer.name}: BN output feeds both a ReLU and another layer"
                v.bn.act = act
            elif act != ACT_NONE:
                # activation on a materialised tensor (ShuffleNetV2: ReLU after Add, models.py:593-595): consumers clamp on
                # load; in backward the gradient w.r.t. the activated value is masked in place before the producer runs
                assert len(self.cons.get(id(layer.inbound[0]), [])) == 1, f"{layer.name}: its input also feeds another layer"
                if self.training:
                    self._emit(ActBwdOp(self, v.store, act, layer.name))
            nv = Val(v.store, v.scale, v.shift, act, v.bn)
            prod = getattr(v.store, "producer", None)
            if prod is not None and v.bn is not None:
                prod.out_val = nv
            setv(nv)
        elif isinstance(layer, K.Add):
            st = self._out_store(layer, out_t.shape)
            op = self._emit(ApplyOp(self, ins[1], ins[0], st, layer.name))
            a = ins[1]
            if (self.training and a.bn is not None and a.store.parent is None and st.parent is None and a.store.ld == st.ld and a.store.c == st.c
                    and len(self.cons.get(id(layer.inbound[1]), [])) == 1 and id(layer.inbound[1]) not in {id(t) for t in self.model.outputs}):
                a.store._grad = st._raw_grad()    # dL/d(projection output) == dL/d(Add output): one buffer
                op.alias_a = True
            setv(Val(st))
        elif isinstance(layer, K.Concatenate):
            self._lower_concat(layer, ins, setv)
        elif isinstance(layer, K.GlobalAveragePooling2D):
            st = self._out_store(layer, out_t.shape)
            self._emit(GapOp(self, self._dense(ins[0], layer.name), st, layer.name))
            setv(Val(st))
        elif isinstance(layer, K.UpSampling2D):
            cons = self.cons.get(id(out_t), [])
            if len(cons) == 1 and isinstance(cons[0], K.Softmax):
                setv(Val(ins[0].store, ins[0].scale, ins[0].shift, ins[0].act, ins[0].bn, lazy_up=layer.size, src=ins[0]))
            else:
                st = self._out_store(layer, out_t.shape)
                self._emit(BilinearOp(self, ins[0], st, layer.size[0], layer.size[1], layer.name))
                # values are already activated; the concat-wide activation is idempotent on them
                setv(Val(st, materialised_act=ins[0].act))
        elif isinstance(layer, K.Softmax):
            self._lower_softmax(layer, ins[0], setv)
        elif isinstance(layer, K.Reshape):
            self._lower_reshape(layer, ins[0], setv)
        elif isinstance(layer, L.DecodeBoxesCentroidsOffsets):
            setv(Val(ins[0].store, decode=layer))
        elif isinstance(layer, L.SegmentationSuppression):
            setv(Val(ins[1].store, suppress_with=ins[0].store))
        elif isinstance(layer, L.NonMaximumSuppression):
            boxes_v, probs_v = ins[0], ins[1]
            out = Store(self, self.batch, layer.max_number_of_boxes_per_sample, 1, 6, layer.name, need_grad=False)
            self.stores.append(out)
            self._emit(DecodeNmsOp(self, boxes_v.store, probs_v.store, boxes_v.meta["decode"], layer, probs_v.meta.get("suppress_with"), out))
            setv(Val(out, nms=layer))
        else:
            self._lower_shufflenet(layer, ins, setv)

    def _lower_conv(self, layer: K.Conv2D, v: Val, setv):
        out_t = layer.outputs[0]
        if "rescale" in v.meta:
            assert layer.kernel_size == (3, 3) and layer.strides == (2, 2), "Rescaling must feed the 3x3 stride-2 stem"
            st = self._out_store(layer, out_t.shape)
            op = self._emit(StemOp(self, layer, v.store, v.meta["rescale"], st))
        elif layer.kernel_size == (1, 1):
            assert layer.strides == (1, 1) and not layer.use_bias
            st = self._out_store(layer, out_t.shape)
            op = self._emit(PwOp(self, layer, "kernel", v, st))
            src = layer.inbound[0]
            # only for wide inputs (the 6x depthwise tensor in front of a project conv); the narrow block-input tensors in front of
            # expand convs take the fused dx+dW kernel instead
            op.fuse_input_bn = (v.bn is not None and v.store.c > layer.filters and v.store.parent is None
                                and len(self.cons.get(id(src), [])) == 1 and id(src) not in {id(t) for t in self.model.outputs})
        elif layer.kernel_size == (3, 3):
            assert layer.strides == (1, 1) and not layer.use_bias and layer.dilation_rate == (1, 1)
            st = self._out_store(layer, out_t.shape)
            op = self._emit(Conv3Op(self, layer, v, st))
            src = layer.inbound[0]
            # the narrow (<= 8 filters) form writes dx from a pointwise GEMM: the producer BN's backward sums can ride there
            op.fuse_input_bn = (v.bn is not None and layer.filters <= 8 and v.store.parent is None and v.store.ld == v.store.c
                                and len(self.cons.get(id(src), [])) == 1 and id(src) not in {id(t) for t in self.model.outputs})
        else:
            raise NotImplementedError(f"{layer.name}: Conv2D {layer.kernel_size} stride {layer.strides}")
        op.out_val = Val(st)
        st.producer = op
        setv(op.out_val)

    def _lower_concat(self, layer: K.Concatenate, ins: List[Val], setv):
        out_t = layer.outputs[0]
        if layer.axis_resolved == 1 and all("reshaped" in v.meta for v in ins):
            # SSD heads: (B, boxes_i, 4) blocks stacked along the anchor axis
            total = out_t.shape[1]
            st = Store(self, self.batch, total, 1, out_t.shape[2], layer.name)
            self.stores.append(st)
            off = 0
            for v in ins:
                self._emit(HeadGatherOp(self, Val(v.store, v.scale, v.shift, v.act, v.bn), st, off, total, f"{layer.name}:{v.store.name}"))
                off += v.meta["reshaped"][0]
            assert off == total
            setv(Val(st, head_concat=True))
            return
        assert layer.axis_resolved == 3, f"{layer.name}: unsupported concat axis"
        parent = self._concat_parts(layer)
        acts = set()
        off = 0
        copied_nonneg = True
        for t, v in zip(layer.inbound, ins):
            c = pad4(t.shape[-1])
            if v.store.parent is parent and v.store.coff == off:
                acts.add(v.meta.get("materialised_act", v.act))
            else:
                # not placed: copy the activated values into the slice (identity affine there)
                sl = parent.slice(off, c, f"{layer.name}[{off}:{off + c}]")
                self._emit(ApplyOp(self, v, None, sl, f"{layer.name}:copy{off}"))
                acts.add(None)
                copied_nonneg = copied_nonneg and _nonneg(v)
            off += c
        real = {a for a in acts if a is not None}
        if None in acts and real == {ACT_RELU} and copied_nonneg:
            # ShuffleNetV2 basic unit (models.py:573-598): the untouched half holds values >= 0 (it comes out of a ReLU'd
            # shuffle / max-pool), so the other half's lazy ReLU is the identity on it and the concat can stay a view
            setv(Val(parent, parent.wide_scale, parent.wide_shift, ACT_RELU))
        elif None in acts:
            # copied slices hold final values: only legal to keep lazy activations if there are none left
            assert not real or real == {ACT_NONE}, f"{layer.name}: cannot mix copied and lazily-activated inputs"
            setv(Val(parent))
        else:
            assert len(real) == 1, f"{layer.name}: concat inputs carry different activations {real}"
            setv(Val(parent, parent.wide_scale, parent.wide_shift, real.pop()))

    def _lower_softmax(self, layer: K.Softmax, v: Val, setv):
        out_t = layer.outputs[0]
        if "lazy_up" in v.meta:
            fy, fx = v.meta["lazy_up"]
            src: Val = v.meta["src"]
            n, h, w, c = (self.batch,) + tuple(out_t.shape[1:])
            prob = None
            if not self.training or self.keep_mask_probabilities:
                prob = Store(self, n, h, w, c, layer.name, need_grad=False)
                self.stores.append(prob)
            op = self._emit(MaskHeadOp(self, src, fy, fx, layer.name, prob))
            self.loss_ops[layer.name] = op
            setv(Val(prob if prob is not None else src.store, mask_head=op))
        elif v.meta.get("head_concat"):
            s = v.store
            st = Store(self, s.n, s.h, s.w, s.c, layer.name, need_grad=False)
            self.stores.append(st)
            self._emit(SoftmaxRowsOp(self, s, st, layer.name))
            setv(Val(st, logits=s))
        else:
            raise NotImplementedError(f"{layer.name}: Softmax on this tensor kind")

    def _lower_shufflenet(self, layer, ins, setv):
        """ShuffleNetV2-only layers (reference models.py:480-652): max-pool, channel split, channel shuffle"""
        out_t = layer.outputs[0]
        if isinstance(layer, K.MaxPooling2D):
            st = self._out_store(layer, out_t.shape)
            self._emit(MaxPoolOp(self, self._dense(ins[0], layer.name), st, layer.name))
            setv(Val(st, nonneg=_nonneg(ins[0])))
        elif isinstance(layer, L.Split):
            v = ins[0]
            assert v.scale is None and v.act == ACT_NONE and layer.axis in (-1, 3), f"{layer.name}: only plain channel splits are lowered"
            if any(t.shape[-1] % 4 != 0 for t in layer.outputs):
                # parts that are not whole 16-byte channel vectors: gather each into its own zero-padded tensor
                vd = self._dense(v, layer.name)
                outs = []
                for k, t in enumerate(layer.outputs):
                    st = Store(self, vd.store.n, vd.store.h, vd.store.w, pad4(t.shape[-1]), f"{layer.name}[{k}]")
                    self.stores.append(st)
                    outs.append(st)
                    self.vals[id(t)] = Val(st, nonneg=_nonneg(v))
                self._emit(SplitGatherOp(self, vd, outs, [int(t.shape[-1]) for t in layer.outputs], layer.name))
                return
            v.store.split_parent = True
            off = 0
            for t in layer.outputs:
                c = t.shape[-1]
                self.vals[id(t)] = Val(v.store.slice(off, c, f"{layer.name}[{off}:{off + c}]"), nonneg=_nonneg(v))   # zero-copy channel view
                off += c
        elif isinstance(layer, K.Permute):
            v = ins[0]
            assert v.meta.get("shuffle_groups") and layer.dims == (1, 2, 4, 3), f"{layer.name}: only the channel-shuffle Permute is lowered"
            setv(Val(v.store, shuffle_groups=v.meta["shuffle_groups"], shuffle_src=v.meta["shuffle_src"], shuffle_permuted=True))
        else:
            raise NotImplementedError(f"layer type {type(layer).__name__} ({layer.name}) is not lowered yet")

    def _lower_reshape(self, layer: K.Reshape, v: Val, setv):
        out_t = layer.outputs[0]
        tgt = tuple(out_t.shape[1:])
        if len(tgt) == 4:                          # (h, w, groups, c/groups): first half of a channel shuffle (models.py:497)
            setv(Val(v.store, v.scale, v.shift, v.act, v.bn, shuffle_groups=tgt[2], shuffle_src=v))
        elif v.meta.get("shuffle_permuted"):       # back to (h, w, c): emit the shuffle (models.py:503)
            st = self._out_store(layer, out_t.shape)
            src = v.meta["shuffle_src"]
            parts = getattr(src.store, "logical_parts", None)
            if parts is not None and any(cl % 4 != 0 for _, cl in parts):
                self._emit(TableShuffleOp(self, src, st, v.meta["shuffle_groups"], layer.name))
            else:
                self._emit(ShuffleOp(self, src, st, v.meta["shuffle_groups"], layer.name))
            setv(Val(st, nonneg=_nonneg(v.meta["shuffle_src"])))
        else:                                      # SSD heads: (B, H, W, boxes*4) -> (B, H*W*boxes, 4), metadata only
            setv(Val(v.store, v.scale, v.shift, v.act, v.bn, reshaped=tgt, **{k: x for k, x in v.meta.items() if k != "reshaped"}))

    keep_mask_probabilities = False

    # ------------------------------------------------------------------ execution
    def set_input(self, images):
        if isinstance(images, H.DeviceBuffer):
            if images.ptr != self.input_store.buf.ptr:
                self.input_store.buf.copy_from(images)
        else:
            a = np.ascontiguousarray(images, dtype=np.float32)
            assert a.shape == (self.batch, self.input_store.h, self.input_store.w, self.input_store.c), f"input shape {a.shape}"
            self.input_store.buf.upload(a)

    def forward(self):
        self._sync_padded("p", to_bucket=False)          # padded copies of the weights <- bucket (no-op for most models)
        self._refresh_wt()                                # transposed copies of the pointwise kernels: one launch
        trunk, det, mask, _ = self._schedule()
        for op in trunk:
            op.fwd()
        if det:
            self.ctx.side(True)              # forks behind the trunk: the detection branch runs beside the mask branch
            for op in det:
                op.fwd()
            self.ctx.side(False)
            for op in mask:
                op.fwd()
            if getattr(self, "_metrics", None):
                self.ctx.join()              # the metric kernels read both branches' outputs on the main stream
        if self.training:
            self._sync_padded("p", to_bucket=True, which="state")   # moving statistics of padded BatchNorms -> bucket

    def _defer_colsums(self):
        """the column sums that fold the weight-gradient partial slabs are recorded during the pass and launched ONCE by the
        join at its end (include/ssdseg.h: ssdseg_colsum_defer); SSDSEG_COLSUM_DEFER=0 keeps one launch per layer (A/B runs,
        bit-identical results)"""
        self.ctx.colsum_defer(os.environ.get("SSDSEG_COLSUM_DEFER", "1") != "0")

    def backward(self):
        self._defer_colsums()
        for s in self.stores:
            s.gwritten = False
        for s in self.stores:
            if s.split_parent and s.need_grad:
                s.grad.zero_()
                s.gwritten = True
        trunk, det, mask, join_before = self._schedule()
        if det:
            self.ctx.side(True)
            for op in reversed(det):
                op.bwd()
            self.ctx.side(False)
            # the main stream has to wait for the END OF THE DETECTION BRANCH twice -- not for the side stream as a whole, which by
            # then also carries the mask branch's weight gradients, queued behind it (a join there stalled the main stream for them)
            wait = self.ctx.join if os.environ.get("SSDSEG_DET_SIDE") == "join" else self.ctx.side_wait_mark      # (A/B: full joins)
            self.ctx.side_mark()
            for op in reversed(mask):
                if id(op) in join_before:
                    wait()                         # this op adds to a gradient the detection branch wrote first
                    join_before = ()
                op.bwd()
            wait()                                 # the trunk's backward reads both branches' gradients
        for op in reversed(trunk):
            op.bwd()
        for s in self.stores:
            s._materialise_pending()
        self.ctx.join()     # weight-gradient kernels on the side stream: done before anyone (optimizer, all-reduce) reads them
        self._sync_padded("g", to_bucket=True)   # gradients of zero-padded weights -> their Keras-shaped slots of the bucket

    def seed_output_grad(self, index: int, g):
        """inject dL/d(output[index]) (bench config 2 / tests): output values are the ACTIVATED tensors"""
        v = self.output_vals[index]
        buf = v.store.grad
        if isinstance(g, H.DeviceBuffer):
            buf.copy_from(g)
        else:
            buf.upload(np.ascontiguousarray(g, np.float32))

    def mark_output_grads_written(self):
        for v in self.output_vals:
            v.store.gwritten = True

    def backward_from_outputs(self):
        self._defer_colsums()
        for s in self.stores:
            s.gwritten = False
        self.mark_output_grads_written()
        for op in reversed(self.ops):
            op.bwd()
        for s in self.stores:
            s._materialise_pending()
        self.ctx.join()
        self._sync_padded("g", to_bucket=True)

    def output(self, index: int) -> np.ndarray:
        """activated value of output `index` as a NumPy array (N, H, W, C) -- for tests / predict"""
        v = self.output_vals[index]
        s = v.store
        if v.scale is None and v.act == ACT_NONE and s.ld == s.c:
            arr = s.buf.download().reshape(s.n, s.h, s.w, s.ld)
        else:
            tmp = self.ctx.empty((s.m, s.c))
            self.ctx.call("ssdseg_bn_apply", v.view(), s.ld, None, 0, tmp, s.c, s.m, s.c)
            arr = tmp.download().reshape(s.n, s.h, s.w, s.c)
        shp = self.model.outputs[index].shape
        if arr.shape[-1] != shp[-1] and arr.ndim == 4 and len(shp) == 4:
            arr = np.ascontiguousarray(arr[..., :shp[-1]])      # zero-padded channels of an odd-width tensor
        return arr.reshape((s.n,) + tuple(shp[1:]))

    # ------------------------------------------------------------------ training step (Keras train_step stand-in)
    def configure_losses(self, loss: dict, loss_weights: dict):
        """bind the compiled losses to the fused loss ops (reference NB03#cell14: cross_entropy(w) / confidence_loss /
        localization_loss, weights 1/1/1; Keras reduces each (B,) loss by its batch mean)"""
        from . import losses as LS
        self._loss_names = []
        b = self.batch
        for t in self.model.outputs:
            name = t.layer.name
            fn = loss.get(name)
            if fn is None:
                continue
            w = float(loss_weights.get(name, 1.0))
            v = self.vals[id(t)]
            if "mask_head" in v.meta:
                if getattr(fn, "loss_kind", None) not in ("cross_entropy", "dice", "dice_square"):
                    raise NotImplementedError(f"loss for output {name}: the mask head trains with ssdseglib.losses.cross_entropy / dice / "
                                              "dice_square")
                op: MaskHeadOp = v.meta["mask_head"]
                op.kind = fn.loss_kind
                op.coef = self.ctx.empty((b, 8)) if op.kind != "cross_entropy" else None
                s = op.logits.store
                op.y_true = self.ctx.empty((b, s.h * op.fy, s.w * op.fx, s.c))
                op.class_weights = (C.c_float * 4)(*[float(x) for x in fn.classes_weights])
                op.loss = self.ctx.empty(b)
                op.loss_scale = w / b
                self._loss_names.append((name, op, "mask"))
            elif fn is LS.confidence_loss or fn is LS.localization_loss:
                det = self.loss_ops.get("det")
                if det is None:
                    labels_v = next(self.vals[id(o)] for o in self.model.outputs if "logits" in self.vals[id(o)].meta)
                    boxes_v = next(self.vals[id(o)] for o in self.model.outputs if self.vals[id(o)].meta.get("head_concat"))
                    det = DetLossOp(self, labels_v.meta["logits"], labels_v.store, boxes_v.store)
                    a = labels_v.store.h * labels_v.store.w
                    det.y_labels = self.ctx.empty((b, a, labels_v.store.c))
                    det.y_boxes = self.ctx.empty((b, a, 4))
                    self.loss_ops["det"] = det
                    self._cur_reach = self.DET_OUTPUTS
                    self._emit(det)
                if fn is LS.confidence_loss:
                    det.w_conf = w
                    self._loss_names.append((name, det, "conf"))
                else:
                    det.w_loc = w
                    self._loss_names.append((name, det, "loc"))
            else:
                raise NotImplementedError(f"loss for output {name}: only the ssdseglib.losses functions are supported")

    def configure_metrics(self, metrics: dict):
        """bind compile(metrics={output: fn | [fn, ...]}) (NB03#cell14) to the device buffers of this engine's step: the
        ssdseglib.metrics callables are evaluated by `ssdseg_metric_*` right after the forward pass, from the same target
        buffers the losses use (needs configure_losses first); Keras naming `<output>_<function name>`"""
        self._metrics = []
        targets = {name: (op, kind) for name, op, kind in self._loss_names}
        for t in self.model.outputs:
            name = t.layer.name
            fns = metrics.get(name)
            if fns is None:
                continue
            for fn in (fns if isinstance(fns, (list, tuple)) else [fns]):
                kind = getattr(fn, "metric_kind", None)
                if kind is None:
                    raise NotImplementedError(f"metric for output {name}: only the ssdseglib.metrics factories are supported")
                if name not in targets:
                    raise NotImplementedError(f"metric for output {name} needs a compiled loss on the same output (it shares its target buffer)")
                op, lkind = targets[name]
                want = {"mask_iou": "mask", "label_accuracy": "conf", "box_iou": "loc"}[kind]
                if want != lkind:
                    raise ValueError(f"metric {fn.__name__} does not fit output {name}")
                self._metrics.append((f"{name}_{fn.__name__}", kind, fn, op, self.ctx.empty(self.batch)))

    def compute_metrics(self):
        """launch the metric kernels on the outputs of the forward pass just run (asynchronous; read by `losses()`)"""
        for key, kind, fn, op, out in getattr(self, "_metrics", []):
            if kind == "mask_iou":
                s = op.logits.store
                self.ctx.call("ssdseg_metric_mask_iou", s.buf, s.n, s.h, s.w, s.c, op.fy, op.fx, 1, op.y_true, fn._cw, out)
            elif kind == "label_accuracy":
                p = op.probs
                self.ctx.call("ssdseg_metric_label_accuracy", op.y_labels, p.buf, p.n, p.h * p.w, p.c, fn._cw, out)
            else:
                p = op.boxes
                self.ctx.call("ssdseg_metric_box_iou", op.y_boxes, p.buf, fn.anchors_on(self.ctx), fn._stds, p.n, p.h * p.w, out)

    def set_targets(self, targets: dict):
        for name, op, kind in self._loss_names:
            y = targets[name]
            dst = op.y_true if kind == "mask" else (op.y_labels if kind == "conf" else op.y_boxes)
            if isinstance(y, H.DeviceBuffer):
                if y.ptr != dst.ptr:
                    dst.copy_from(y)
            else:
                dst.upload(np.ascontiguousarray(y, np.float32))

    def losses(self) -> Dict[str, float]:
        """batch means of the per-sample losses, Keras naming (`loss`, `<output>_loss`)"""
        out, total = {}, 0.0
        for name, op, kind in self._loss_names:
            buf = op.loss if kind == "mask" else (op.conf_loss if kind == "conf" else op.loc_loss)
            w = op.loss_scale * self.batch if kind == "mask" else (op.w_conf if kind == "conf" else op.w_loc)
            v = float(buf.download().mean())
            out[f"{name}_loss"] = v
            total += w * v
        out["loss"] = total
        for key, kind, fn, op, buf in getattr(self, "_metrics", []):
            out[key] = float(buf.download().mean())        # batch mean, NaN-propagating like Keras (quirk Q10)
        return out

    def train_step(self, images=None, targets=None, optimizer=None, allreduce=None, world=1):
        if images is not None:
            self.set_input(images)
        if targets is not None:
            self.set_targets(targets)
        self.forward()
        self.compute_metrics()
        self.backward()
        if allreduce is not None:
            allreduce()
        o = optimizer
        self.adam_step(lr=o.learning_rate, beta1=o.beta_1, beta2=o.beta_2, eps=o.epsilon, grad_scale=1.0 / world) if o is not None \
            else self.adam_step(grad_scale=1.0 / world)

    def adam_step(self, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        P = self.P
        P["step"] = P.get("step", 0) + 1          # shared by every engine (batch size) of the same model
        self.ctx.call("ssdseg_adam_step", P["params"], P["grads"], P["adam_m"], P["adam_v"], C.c_size_t(P["n_tr"]), lr, beta1, beta2, eps,
                      P["step"], grad_scale)


_default_ctx: Optional[H.Context] = None


def default_context() -> H.Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = H.Context(0)
    return _default_ctx


def set_default_context(ctx: H.Context):
    global _default_ctx
    _default_ctx = ctx


def engine_for(model: K.Model, batch_size: int, training: bool) -> Engine:
    cache = model.__dict__.setdefault("_engines", {})
    key = (int(batch_size), bool(training))
    if key not in cache:
        eng = Engine(model, batch_size, training)
        if training:
            comp = model._compiled
            if comp is None:
                raise RuntimeError("call model.compile(optimizer=..., loss=...) before fit / train_on_batch")
            eng.configure_losses(comp["loss"], comp["loss_weights"])
            eng.configure_metrics(comp.get("metrics") or {})
        cache[key] = eng
    return cache[key]


def eval_engine_for(model: K.Model, batch_size: int) -> Engine:
    """inference-mode engine that also evaluates the compiled losses (validation_data of fit)"""
    cache = model.__dict__.setdefault("_engines", {})
    key = (int(batch_size), "eval")
    if key not in cache:
        eng = Engine(model, batch_size, False)
        eng.configure_losses(model._compiled["loss"], model._compiled["loss_weights"])
        eng.configure_metrics(model._compiled.get("metrics") or {})
        cache[key] = eng
    return cache[key]


# ---------------------------------------------------------------------------------------------- Keras-like entry points
def _postprocess(model: K.Model, eng: Engine, index: int) -> np.ndarray:
    out = eng.output(index)
    v = eng.output_vals[index]
    nms = v.meta.get("nms")
    if nms is not None and nms.suppress_background_boxes:
        # quirk Q7 (reference layers.py:165-166): boolean_mask drops the batch axis
        out = out.reshape(-1, 6)
        out = out[out[:, 0] > 0]
    return out


def run_forward(model: K.Model, x, training: bool = False) -> List[np.ndarray]:
    """model(x, training=False) -> list of output arrays (NB03#cell31)"""
    x = np.asarray(x, np.float32)
    if x.ndim == 3:
        x = x[None]
    eng = engine_for(model, x.shape[0], False)
    eng.set_input(x)
    eng.forward()
    return [_postprocess(model, eng, i) for i in range(len(model.outputs))]


def _batches(data):
    """accepts an iterable of batches: x, (x,), (x, y) with y a dict keyed by output name"""
    for item in data:
        if isinstance(item, (tuple, list)):
            yield item[0], (item[1] if len(item) > 1 else None)
        else:
            yield item, None


def run_predict(model: K.Model, data) -> List[np.ndarray]:
    """model.predict(dataset) -> outputs concatenated over the batches (NB03#cell25)"""
    if isinstance(data, np.ndarray):
        data = [data[i:i + 16] for i in range(0, data.shape[0], 16)]
    chunks: List[List[np.ndarray]] = []
    for x, _ in _batches(data):
        chunks.append(run_forward(model, x))
    return [np.concatenate([c[i] for c in chunks], axis=0) for i in range(len(model.outputs))]


def run_train_on_batch(model: K.Model, x, y=None) -> Dict[str, float]:
    if _is_compact(x):
        eng = engine_for(model, len(x), True)
        ld = _compact_loader(eng, x)
        ld.stage(x)
        ld.consume()
        eng.train_step(optimizer=model._compiled.get("optimizer"))
        return eng.losses()
    x = np.asarray(x, np.float32)
    eng = engine_for(model, x.shape[0], True)
    eng.train_step(x, y, optimizer=model._compiled.get("optimizer"))
    return eng.losses()


class _BatchStager:
    """Hands the NEXT batch to the device while the current step runs (fit's input side).

    The host arrays are uploaded on the context's copy stream into device staging buffers while the GPU computes the current
    step (the host thread blocks in the copy -- the runtime pipelines pageable memory through its own pinned buffers at
    ~45 GB/s -- but it has nothing else to do until the step's losses are ready); at the start of the step that uses them
    the main stream waits for that upload and copies staging -> the engine's input / target buffers (device to device,
    ~0.3 ms for 285 MB).  The engine's own buffers cannot be the upload target: the running step still reads them (the mask
    targets until the end of the backward pass).  A first version copied into pinned buffers of its own: that host memcpy
    (285 MB, one thread) took as long as the step and bought nothing."""

    def __init__(self, eng: "Engine"):
        self.eng = eng
        self.ctx = eng.ctx
        self.dst = [("__input__", eng.input_store.buf)]
        for name, op, kind in eng._loss_names:
            self.dst.append((name, op.y_true if kind == "mask" else (op.y_labels if kind == "conf" else op.y_boxes)))
        self.staging = [self.ctx.empty(d.shape) for _, d in self.dst]
        self.staged = False
        self._keep = None      # the host arrays of the upload in flight

    def stage(self, x, y) -> bool:
        """host side, while the GPU is busy with the current step: fill the pinned buffers, enqueue the uploads"""
        arrays = [x] + [y[name] for name, _ in self.dst[1:]]
        if any(isinstance(a, H.DeviceBuffer) for a in arrays) or any(int(np.prod(np.shape(a))) != st.size for a, st in zip(arrays, self.staging)):
            return False
        arrays = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]   # no copy for float32 C-contiguous batches
        self.ctx.upload_sync()
        for a, st in zip(arrays, self.staging):
            self.ctx.upload_async(st, a, after_fence=True)   # behind the last consume(), under the step queued after it
        self._keep = arrays
        self.staged = True
        return True

    def consume(self):
        """device side, at the start of the step: staging -> the engine's buffers"""
        assert self.staged
        self.ctx.upload_join()
        for (_, d), st in zip(self.dst, self.staging):
            d.copy_from(st)
        self.ctx.upload_fence()                     # from here on the staging buffers may be overwritten
        self.staged = False


class _CompactLoader:
    """A `datacoder.CompactBatch` into the engine's input / target buffers (reference datacoder.py:302-347 == csrc/inputs.hip +
    ssdseg_encode_targets).  stage(): 39 MB of uint8 pixels / class indices / ground-truth rows go up on the copy stream (under
    the running step when fit overlaps); consume(): the main stream waits for them and expands -- float32 image, one-hot mask,
    both mirrored where flagged, mirrored boxes, encoded anchors -- straight into the buffers the step reads."""

    GMAX = 64       # ground-truth rows per image the encode kernel holds in LDS (boxes.hip)

    def __init__(self, eng: "Engine", encoder):
        self.eng, self.ctx, self.enc = eng, eng.ctx, encoder
        ctx, b, ins = eng.ctx, eng.batch, eng.input_store
        ops = {kind: op for _, op, kind in eng._loss_names}
        self.mask_op, self.det = ops.get("mask"), ops.get("conf") or ops.get("loc")
        self.img = ctx.empty((b, ins.h, ins.w, 3), np.uint8)
        self.midx = ctx.empty((b, ins.h, ins.w), np.uint8)
        self.flip = ctx.empty(b, np.uint8)
        self.gt = ctx.empty((b, self.GMAX, 5))
        self.cnt = ctx.empty(b, np.int32)
        corners = np.stack([encoder.xmin_boxes_default, encoder.ymin_boxes_default, encoder.xmax_boxes_default, encoder.ymax_boxes_default], axis=1)
        self.anchors = ctx.array(corners.astype(np.float32))
        if self.det is not None and self.anchors.shape[0] != self.det.y_boxes.shape[1]:
            raise ValueError(f"encoder has {self.anchors.shape[0]} default boxes, the model's heads {self.det.y_boxes.shape[1]}")
        # ssdseg_encode_targets writes (batch, anchors, encoder.num_classes) floats into the labels target and the mask is
        # expanded to encoder.num_classes planes (reference datacoder.py:332): both buffers are sized by the MODEL's channels
        if self.det is not None and int(encoder.num_classes) != self.det.y_labels.shape[-1]:
            raise ValueError(f"encoder.num_classes = {encoder.num_classes}, the model's labels head has {self.det.y_labels.shape[-1]} classes")
        if self.mask_op is not None and int(encoder.num_classes) != self.mask_op.y_true.shape[-1]:
            raise ValueError(f"encoder.num_classes = {encoder.num_classes}, the model's mask head has {self.mask_op.y_true.shape[-1]} classes")
        self.staged = None
        self._keep = None

    def stage(self, cb) -> None:
        b, ins = self.eng.batch, self.eng.input_store
        if len(cb) != b or cb.images.shape[1:3] != (ins.h, ins.w):
            raise ValueError(f"compact batch {cb.images.shape} for an engine of batch {b}, {ins.h}x{ins.w}")
        if max(g.shape[0] for g in cb.ground_truth) > self.GMAX:
            raise ValueError(f"more than {self.GMAX} ground-truth boxes in one image")
        gt = np.zeros((b, self.GMAX, 5), np.float32)
        cnt = np.zeros(b, np.int32)
        for i, g in enumerate(cb.ground_truth):
            gt[i, :g.shape[0]] = g
            cnt[i] = g.shape[0]
        flip = cb.flip if cb.flip is not None else np.zeros(b, np.uint8)
        self.ctx.upload_sync()
        pairs = [(self.img, cb.images), (self.midx, cb.mask_index), (self.gt, gt), (self.cnt, cnt), (self.flip, flip)]
        for dst, src in pairs:
            self.ctx.upload_async(dst, src, after_fence=True)
        self._keep = [src for _, src in pairs]
        self.staged = bool(flip.any())

    def consume(self) -> None:
        assert self.staged is not None
        ctx, b, ins, enc = self.ctx, self.eng.batch, self.eng.input_store, self.enc
        ctx.upload_join()
        flip = self.flip if self.staged else None
        c = self.mask_op.y_true.shape[-1] if self.mask_op is not None else 1
        ctx.call("ssdseg_expand_inputs", self.img, self.midx if self.mask_op is not None else None, flip, ins.buf,
                 self.mask_op.y_true if self.mask_op is not None else None, b, ins.h, ins.w, c)
        if self.det is not None:
            if flip is not None:
                ctx.call("ssdseg_flip_gt_boxes", self.gt, self.cnt, flip, b, self.GMAX, float(ins.w))
            ctx.call("ssdseg_encode_targets", self.anchors, self.anchors.shape[0], self.gt, self.cnt, b, self.GMAX, enc.num_classes,
                     float(enc.iou_threshold), (C.c_float * 4)(*enc._stds), self.det.y_labels, self.det.y_boxes, None)
        ctx.upload_fence()                          # from here on the compact staging buffers may be overwritten
        self.staged = None


def _compact_loader(eng: "Engine", cb) -> _CompactLoader:
    ld = eng.__dict__.get("_compact_loader")
    if ld is None or ld.enc is not cb.encoder:
        ld = eng.__dict__["_compact_loader"] = _CompactLoader(eng, cb.encoder)
    return ld


def _is_compact(x) -> bool:
    return type(x).__name__ == "CompactBatch"


class History:
    def __init__(self):
        self.history: Dict[str, List[float]] = {}
        self.epoch: List[int] = []


def run_fit(model: K.Model, data, epochs=1, validation_data=None, verbose=0) -> History:
    """model.fit(ds, epochs, validation_data, verbose) (NB03#cell16): the last partial batch is kept (its own batch
    statistics and mining pool); per-epoch means weighted by batch size, Keras history keys."""
    hist = History()
    overlap = os.environ.get("SSDSEG_FIT_OVERLAP", "1") != "0"
    for epoch in range(epochs):
        sums: Dict[str, float] = {}
        seen = 0
        it = iter(_batches(data))
        cur = next(it, None)
        staged_for = None      # id of the batch whose upload is in flight, and its stager
        while cur is not None:
            x, y = cur
            n = len(x) if _is_compact(x) else (int(np.shape(x)[0]) if not isinstance(x, H.DeviceBuffer) else x.shape[0])
            eng = engine_for(model, n, True)
            if _is_compact(x):
                ld = _compact_loader(eng, x)
                if not (staged_for is not None and staged_for[0] is cur):
                    ld.stage(x)
                ld.consume()                        # expansion + anchor encoding on the device
                eng.train_step(optimizer=model._compiled.get("optimizer"))
            elif staged_for is not None and staged_for[0] is cur:
                staged_for[1].consume()
                eng.train_step(optimizer=model._compiled.get("optimizer"))
            else:
                eng.train_step(np.asarray(x, np.float32) if not isinstance(x, H.DeviceBuffer) else x, y, optimizer=model._compiled.get("optimizer"))
            # the step is queued; everything below runs on the host while the GPU works on it
            nxt = next(it, None)
            staged_for = None
            if overlap and nxt is not None and _is_compact(nxt[0]) and _is_compact(x) and len(nxt[0]) == n and nxt[0].encoder is x.encoder:
                ld = _compact_loader(eng, nxt[0])
                ld.stage(nxt[0])                    # 39 MB on the copy stream, under the step just queued
                staged_for = (nxt, ld)
            elif overlap and nxt is not None and not _is_compact(nxt[0]) and nxt[1] is not None and not isinstance(nxt[0], H.DeviceBuffer) and int(np.shape(nxt[0])[0]) == n:
                stager = eng.__dict__.setdefault("_stager", None) or _BatchStager(eng)
                eng._stager = stager
                if stager.stage(nxt[0], nxt[1]):
                    staged_for = (nxt, stager)
            logs = eng.losses()
            for k, v in logs.items():
                sums[k] = sums.get(k, 0.0) + v * n
            seen += n
            cur = nxt
        logs = {k: v / max(seen, 1) for k, v in sums.items()}
        if validation_data is not None:
            vs: Dict[str, float] = {}
            vseen = 0
            for x, y in _batches(validation_data):
                x = np.asarray(x, np.float32)
                eng = eval_engine_for(model, x.shape[0])   # moving statistics, no gradient buffers (Keras test_step)
                eng.set_input(x)
                eng.set_targets(y)
                eng.forward()
                eng.compute_metrics()
                for k, v in eng.losses().items():
                    vs[k] = vs.get(k, 0.0) + v * x.shape[0]
                vseen += x.shape[0]
            logs.update({f"val_{k}": v / max(vseen, 1) for k, v in vs.items()})
        for k, v in logs.items():
            hist.history.setdefault(k, []).append(v)
        hist.epoch.append(epoch)
        if verbose:
            print(f"Epoch {epoch + 1}/{epochs} - " + " - ".join(f"{k}: {v:.4f}" for k, v in logs.items()))
    return hist
