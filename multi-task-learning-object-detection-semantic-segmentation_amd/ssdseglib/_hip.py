"""ctypes binding of libssdseg_hip.so (C-ABI: include/ssdseg.h) -- the only door from the Python host to the GPU.

No torch, no fallbacks: if the shared library is missing or a call fails this module raises.  Device memory is
owned by `DeviceBuffer` objects (hipMalloc/hipFree through the C-ABI), or borrowed from a raw pointer.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this driver only does dmabuf IPC (RCCL across processes needs it)

_LIB_NAME = "libssdseg_hip.so"
_lib = None

ACT_NONE, ACT_RELU, ACT_RELU6, ACT_ZERO = 0, 1, 2, 3


class SsdsegError(RuntimeError):
    pass


class ViewStruct(C.Structure):
    _fields_ = [("x", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("act", C.c_int32), ("_pad", C.c_int32)]


class GViewStruct(C.Structure):
    _fields_ = [("g", C.c_void_p), ("y", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("k1", C.c_void_p), ("k0", C.c_void_p), ("act", C.c_int32), ("_pad", C.c_int32)]


_vp, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
_VP = C.POINTER(ViewStruct)
_GP = C.POINTER(GViewStruct)
_ip = C.POINTER(C.c_int)

# name -> argtypes (restype is always int, except the two below)
_SIGNATURES = {
    "ssdseg_device_count": [_ip],
    "ssdseg_ctx_create": [_i, _vp, C.POINTER(_vp)],
    "ssdseg_ctx_destroy": [_vp],
    "ssdseg_ctx_sync": [_vp],
    "ssdseg_ctx_join": [_vp],
    "ssdseg_ctx_side": [_vp, _i],
    "ssdseg_ctx_side_enable": [_vp, _i],
    "ssdseg_ctx_side_mark": [_vp],
    "ssdseg_ctx_side_wait_mark": [_vp],
    "ssdseg_colsum_defer": [_vp, _i],
    "ssdseg_ctx_reserve": [_vp, _sz],
    "ssdseg_ctx_device_name": [_vp, C.c_char_p, _sz],
    "ssdseg_malloc": [_vp, _sz, C.POINTER(_vp)],
    "ssdseg_free": [_vp, _vp],
    "ssdseg_memcpy_h2d": [_vp, _vp, _vp, _sz],
    "ssdseg_memcpy_d2h": [_vp, _vp, _vp, _sz],
    "ssdseg_memcpy_d2d": [_vp, _vp, _vp, _sz],
    "ssdseg_memset": [_vp, _vp, _i, _sz],
    "ssdseg_event_create": [_vp, C.POINTER(_vp)],
    "ssdseg_event_destroy": [_vp, _vp],
    "ssdseg_event_record": [_vp, _vp],
    "ssdseg_event_elapsed_ms": [_vp, _vp, _vp, C.POINTER(_f)],
    "ssdseg_timing_enable": [_vp, _i],
    "ssdseg_timing_reset": [_vp],
    "ssdseg_timing_filter": [_vp, C.c_char_p],
    "ssdseg_timing_report": [_vp, C.c_char_p, _sz],
    "ssdseg_host_alloc": [_vp, _sz, C.POINTER(_vp)],
    "ssdseg_host_free": [_vp, _vp],
    "ssdseg_upload_fence": [_vp],
    "ssdseg_upload_async": [_vp, _vp, _vp, _sz, _i],
    "ssdseg_upload_join": [_vp],
    "ssdseg_upload_sync": [_vp],
    "ssdseg_comm_unique_id": [_vp, _sz],
    "ssdseg_comm_init_rank": [_vp, _vp, _sz, _i, _i],
    "ssdseg_comm_destroy": [_vp],
    "ssdseg_comm_info": [_vp, _ip, _ip],
    "ssdseg_allreduce_grads": [_vp, _vp, _sz, _vp, _sz],
    "ssdseg_allreduce": [_vp, _vp, _sz, _i, _i],
    "ssdseg_broadcast": [_vp, _vp, _sz, _i],
    "ssdseg_stem_conv_parts": [_i, _i, _i, _i, _ip],
    "ssdseg_stem_conv_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp],
    "ssdseg_stem_conv_bwd_weight": [_vp, _vp, _GP, _vp, _vp, _i, _i, _i, _i, _i, _f, _f],
    "ssdseg_dwconv_parts": [_i, _i, _i, _i, _i, _i, _ip],
    "ssdseg_dwconv_fwd": [_vp, _VP, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "ssdseg_dwconv_bwd": [_vp, _VP, _vp, _GP, _vp, _vp, _i, _i, _i, _i, _i, _i, _i],
    "ssdseg_dwconv_bwd_bn": [_vp, _VP, _vp, _GP, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "ssdseg_pwconv_parts": [_i, _i, _ip],
    "ssdseg_pwconv_fwd": [_vp, _VP, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "ssdseg_pwconv_wt_floats": [_i, _i, _i, _i, _ip],
    "ssdseg_transpose_batch": [_vp, _vp, _i, _i, C.c_longlong],
    "ssdseg_pwconv_fwd_wt": [_vp, _VP, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ssdseg_pwconv_bwd_data": [_vp, _GP, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i],
    "ssdseg_pwconv_bwd_weight": [_vp, _VP, _i, _GP, _i, _vp, _i, _i, _i],
    "ssdseg_pwconv_bwd": [_vp, _VP, _i, _GP, _i, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _i, _i],
    "ssdseg_pwconv_bwd_bn": [_vp, _VP, _i, _GP, _i, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "ssdseg_conv3x3_parts": [_i, _i, _i, _i, _i, _ip],
    "ssdseg_conv3x3_fwd": [_vp, _VP, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ssdseg_conv3x3_saved_floats": [_i, _i, _i, _i, _i, C.POINTER(C.c_longlong)],
    "ssdseg_conv3x3_fwd_saved": [_vp, _VP, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "ssdseg_conv3x3_fwd_saved_from": [_vp, _VP, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i],
    "ssdseg_conv3x3_bwd_weight_saved": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i],
    "ssdseg_conv3x3_bwd_data": [_vp, _GP, _vp, _vp, _i, _i, _i, _i, _i, _i, _i],
    "ssdseg_conv3x3_bwd_data_bn": [_vp, _VP, _GP, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "ssdseg_conv3x3_bwd_weight": [_vp, _VP, _i, _GP, _vp, _i, _i, _i, _i, _i],
    "ssdseg_bn_finalize": [_vp, _vp, _i, _i, _d, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i],
    "ssdseg_channel_stats_parts": [_i, _i, _ip],
    "ssdseg_channel_stats": [_vp, _vp, _i, _i, _i, _vp],
    "ssdseg_bn_apply": [_vp, _VP, _i, _VP, _i, _vp, _i, _i, _i],
    "ssdseg_bn_bwd_reduce": [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "ssdseg_axpby": [_vp, _vp, _i, _vp, _i, _i, _i, _f, _f],
    "ssdseg_gview_materialize": [_vp, _GP, _i, _i, _i],
    "ssdseg_gap_fwd": [_vp, _VP, _vp, _i, _i, _i],
    "ssdseg_gap_bwd": [_vp, _vp, _vp, _i, _i, _i, _i],
    "ssdseg_bilinear_fwd": [_vp, _VP, _i, _vp, _i, _i, _i, _i, _i, _i, _i],
    "ssdseg_bilinear_fwd_padded": [_vp, _VP, _i, _vp, _i, _i, _i, _i, _i, _i, _i],
    "ssdseg_bilinear_bwd": [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i],
    "ssdseg_mask_head_fwd": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, C.POINTER(_f), _vp, _vp],
    "ssdseg_mask_head_bwd": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, C.POINTER(_f), _f, _vp],
    "ssdseg_mask_head_fwd_dice": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, C.POINTER(_f), _i, _vp, _vp, _vp],
    "ssdseg_mask_head_bwd_dice": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _f, _vp],
    "ssdseg_head_gather": [_vp, _VP, _vp, _i, _i, _i, _i, _i, _i],
    "ssdseg_softmax_rows": [_vp, _VP, _vp, _i, _i],
    "ssdseg_det_loss": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp],
    "ssdseg_topk_mask": [_vp, _vp, _i, _i, _vp],
    "ssdseg_dice_loss": [_vp, _vp, _vp, _i, _i, _i, C.POINTER(_f), _i, _vp],
    "ssdseg_act_bwd": [_vp, _vp, _i, _vp, _i, _i, _i, _i],
    "ssdseg_channel_gather": [_vp, _VP, _i, _vp, _i, C.c_longlong, _i, _vp, _i],
    "ssdseg_copy2d": [_vp, _vp, _i, _vp, _i, _i, _i],
    "ssdseg_copy2d_batch": [_vp, _vp, _i, _i, C.c_longlong],
    "ssdseg_metric_mask_iou": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, C.POINTER(_f), _vp],
    "ssdseg_metric_label_accuracy": [_vp, _vp, _vp, _i, _i, _i, C.POINTER(_f), _vp],
    "ssdseg_metric_box_iou": [_vp, _vp, _vp, _vp, C.POINTER(_f), _i, _i, _vp],
    "ssdseg_expand_inputs": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "ssdseg_flip_gt_boxes": [_vp, _vp, _vp, _vp, _i, _i, _f],
    "ssdseg_encode_targets": [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _f, C.POINTER(_f), _vp, _vp, _vp],
    "ssdseg_decode_boxes": [_vp, _vp, _vp, _i, _i, C.POINTER(_f), _vp],
    "ssdseg_combined_nms": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp],
    "ssdseg_seg_suppress": [_vp, _vp, _i, _i, _vp, _i, _vp],
    "ssdseg_adam_step": [_vp, _vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _i, _f],
    "ssdseg_maxpool3x3s2_fwd": [_vp, _VP, _vp, _i, _i, _i, _i],
    "ssdseg_maxpool3x3s2_bwd": [_vp, _VP, _vp, _vp, _i, _i, _i, _i],
    "ssdseg_channel_shuffle": [_vp, _VP, _i, _vp, _i, _i, _i, _i, _i],
}


def library_path() -> str:
    """the in-tree library; SSDSEG_LIB names another build of the SAME sources (A/B experiments of scripts/dbg: e.g. a kernel
    compiled with a different store form) -- never a fallback: a missing file is an error either way"""
    return os.environ.get("SSDSEG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load_library():
    """dlopen the in-tree shared library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise SsdsegError(f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' "
                          f"or `make -C {os.path.dirname(path)}/../csrc`)")
    lib = C.CDLL(path)
    lib.ssdseg_last_error.restype = C.c_char_p
    lib.ssdseg_last_error.argtypes = []
    lib.ssdseg_version.restype = C.c_int
    lib.ssdseg_version.argtypes = []
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            continue  # exported-symbol completeness is asserted by tests/test_cabi_symbols.py
        fn.restype = C.c_int
        fn.argtypes = argtypes
    _lib = lib
    return lib


def declared_symbols() -> Sequence[str]:
    return ["ssdseg_last_error", "ssdseg_version"] + list(_SIGNATURES)


def _check(rc: int, what: str):
    if rc != 0:
        msg = load_library().ssdseg_last_error().decode("utf-8", "replace")
        raise SsdsegError(f"{what} failed with code {rc}: {msg}")


def _ptr(x) -> Optional[int]:
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    return int(x)


class PinnedBuffer:
    """Page-locked host memory (hipHostMalloc) exposed as a NumPy array: the source of overlapped uploads."""

    def __init__(self, ctx: "Context", shape, dtype=np.float32):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        out = C.c_void_p()
        _check(ctx.lib.ssdseg_host_alloc(ctx.handle, max(self.nbytes, 4), C.byref(out)), "ssdseg_host_alloc")
        self.ptr = out.value
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self.ptr), dtype=self.dtype).reshape(self.shape)

    def __del__(self):
        try:
            if getattr(self, "ptr", None) and self.ctx.handle:
                self.array = None
                self.ctx.lib.ssdseg_host_free(self.ctx.handle, C.c_void_p(self.ptr))
        except Exception:
            pass
        self.ptr = None


class DeviceBuffer:
    """A dense fp32/int32/uint8 array in HBM.  `view(offset_elems, shape)` aliases a sub-range (no copy)."""

    def __init__(self, ctx: "Context", shape, dtype=np.float32, ptr: Optional[int] = None, owner=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        self._owner = owner          # keeps the parent allocation alive for views / borrowed memory
        self._owns = ptr is None
        if ptr is None:
            out = C.c_void_p()
            _check(ctx.lib.ssdseg_malloc(ctx.handle, max(self.nbytes, 4), C.byref(out)), "ssdseg_malloc")
            self.ptr = out.value
            ctx.allocated_bytes += self.nbytes
        else:
            self.ptr = int(ptr)

    def __del__(self):
        try:
            if getattr(self, "_owns", False) and self.ptr and self.ctx.handle:
                self.ctx.lib.ssdseg_free(self.ctx.handle, C.c_void_p(self.ptr))
                self.ctx.allocated_bytes -= self.nbytes
        except Exception:
            pass
        self.ptr = None

    def view(self, offset_elems: int, shape, dtype=None) -> "DeviceBuffer":
        dt = self.dtype if dtype is None else np.dtype(dtype)
        v = DeviceBuffer(self.ctx, shape, dt, ptr=self.ptr + int(offset_elems) * self.dtype.itemsize, owner=self)
        assert int(offset_elems) * self.dtype.itemsize + v.nbytes <= self.nbytes, "view out of range"
        return v

    def reshape(self, shape) -> "DeviceBuffer":
        v = DeviceBuffer(self.ctx, shape, self.dtype, ptr=self.ptr, owner=self)
        assert v.size == self.size
        return v

    def upload(self, array) -> "DeviceBuffer":
        a = np.ascontiguousarray(array, dtype=self.dtype)
        assert a.size == self.size, f"upload size mismatch {a.shape} vs {self.shape}"
        _check(self.ctx.lib.ssdseg_memcpy_h2d(self.ctx.handle, self.ptr, a.ctypes.data, self.nbytes), "ssdseg_memcpy_h2d")
        return self

    def download(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        _check(self.ctx.lib.ssdseg_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes), "ssdseg_memcpy_d2h")
        return out

    def zero_(self) -> "DeviceBuffer":
        _check(self.ctx.lib.ssdseg_memset(self.ctx.handle, self.ptr, 0, self.nbytes), "ssdseg_memset")
        return self

    def copy_from(self, other: "DeviceBuffer") -> "DeviceBuffer":
        assert other.nbytes == self.nbytes
        _check(self.ctx.lib.ssdseg_memcpy_d2d(self.ctx.handle, self.ptr, other.ptr, self.nbytes), "ssdseg_memcpy_d2d")
        return self


class Event:
    def __init__(self, ctx: "Context"):
        self.ctx = ctx
        out = C.c_void_p()
        _check(ctx.lib.ssdseg_event_create(ctx.handle, C.byref(out)), "ssdseg_event_create")
        self.handle = out.value

    def record(self):
        _check(self.ctx.lib.ssdseg_event_record(self.ctx.handle, self.handle), "ssdseg_event_record")
        return self

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float()
        _check(self.ctx.lib.ssdseg_event_elapsed_ms(self.ctx.handle, self.handle, stop.handle, C.byref(ms)), "ssdseg_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle and self.ctx.handle:
                self.ctx.lib.ssdseg_event_destroy(self.ctx.handle, self.handle)
        except Exception:
            pass


def view(x, scale=None, shift=None, act=ACT_NONE) -> ViewStruct:
    v = ViewStruct(_ptr(x), _ptr(scale), _ptr(shift), int(act), 0)
    v._keep = (x, scale, shift)      # the struct only holds raw pointers: keep the buffers alive with it
    return v


def gview(g, y=None, scale=None, shift=None, k1=None, k0=None, act=ACT_NONE) -> GViewStruct:
    v = GViewStruct(_ptr(g), _ptr(y), _ptr(scale), _ptr(shift), _ptr(k1), _ptr(k0), int(act), 0)
    v._keep = (g, y, scale, shift, k1, k0)
    return v


class Context:
    """One (device, stream) execution context.  `stream` may be a raw hipStream_t (int) to borrow."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = load_library()
        self.handle = None
        self.allocated_bytes = 0
        out = C.c_void_p()
        _check(self.lib.ssdseg_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(out)), "ssdseg_ctx_create")
        self.handle = out.value
        self.device = int(device)

    def close(self):
        if self.handle:
            self.lib.ssdseg_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        pass  # buffers may outlive us during interpreter shutdown; the process exit frees the device

    # ---- memory
    def empty(self, shape, dtype=np.float32) -> DeviceBuffer:
        return DeviceBuffer(self, shape, dtype)

    def zeros(self, shape, dtype=np.float32) -> DeviceBuffer:
        return DeviceBuffer(self, shape, dtype).zero_()

    def array(self, a, dtype=None) -> DeviceBuffer:
        a = np.asarray(a)
        dt = a.dtype if dtype is None else np.dtype(dtype)
        return DeviceBuffer(self, a.shape, dt).upload(a)

    def borrow(self, ptr: int, shape, dtype=np.float32, owner=None) -> DeviceBuffer:
        return DeviceBuffer(self, shape, dtype, ptr=ptr, owner=owner)

    def sync(self):
        _check(self.lib.ssdseg_ctx_sync(self.handle), "ssdseg_ctx_sync")

    def join(self):
        """main stream waits for the side-stream work queued so far (weight gradients)"""
        _check(self.lib.ssdseg_ctx_join(self.handle), "ssdseg_ctx_join")

    def side(self, on: bool):
        """route the following launches to the side stream (weight gradients) / back to the ctx stream"""
        _check(self.lib.ssdseg_ctx_side(self.handle, 1 if on else 0), "ssdseg_ctx_side")

    def side_mark(self):
        """remember the current end of the side stream (include/ssdseg.h)"""
        _check(self.lib.ssdseg_ctx_side_mark(self.handle), "ssdseg_ctx_side_mark")

    def side_wait_mark(self):
        """the ctx stream waits for the marked point of the side stream -- not for what was queued there later"""
        _check(self.lib.ssdseg_ctx_side_wait_mark(self.handle), "ssdseg_ctx_side_wait_mark")

    def side_enable(self, enabled: bool):
        _check(self.lib.ssdseg_ctx_side_enable(self.handle, 1 if enabled else 0), "ssdseg_ctx_side_enable")

    def colsum_defer(self, enabled: bool):
        """record the column sums of weight-gradient slabs and fold them with one launch at the next join (include/ssdseg.h)"""
        _check(self.lib.ssdseg_colsum_defer(self.handle, 1 if enabled else 0), "ssdseg_colsum_defer")

    def reserve(self, nbytes: int):
        _check(self.lib.ssdseg_ctx_reserve(self.handle, int(nbytes)), "ssdseg_ctx_reserve")

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        _check(self.lib.ssdseg_ctx_device_name(self.handle, buf, 256), "ssdseg_ctx_device_name")
        return buf.value.decode()

    def event(self) -> Event:
        return Event(self)

    # ---- per-kernel HIP-event timing (bench.py roofline leg)
    def timing(self, enable: bool):
        _check(self.lib.ssdseg_timing_enable(self.handle, 1 if enable else 0), "ssdseg_timing_enable")

    def timing_filter(self, kernel: Optional[str]):
        _check(self.lib.ssdseg_timing_filter(self.handle, kernel.encode() if kernel else None), "ssdseg_timing_filter")

    def timing_reset(self):
        _check(self.lib.ssdseg_timing_reset(self.handle), "ssdseg_timing_reset")

    def timing_report(self):
        """-> {kernel symbol: dict(count, ms, bytes, flops, view_bytes)}: `bytes` = SURVEY.md 8(d) algorithmic bytes, `view_bytes` =
        the second tensor of BatchNorm-backward gradient views read on top of them"""
        buf = C.create_string_buffer(1 << 16)
        _check(self.lib.ssdseg_timing_report(self.handle, buf, len(buf)), "ssdseg_timing_report")
        out = {}
        for line in buf.value.decode().splitlines():
            name, count, ms, nbytes, flops, vbytes = line.split("\t")
            out[name] = dict(count=int(count), ms=float(ms), bytes=float(nbytes), flops=float(flops), view_bytes=float(vbytes))
        return out

    # ---- overlapped uploads (copy stream)
    def upload_async(self, dst: "DeviceBuffer", src, after_fence: bool = True):
        """src: a PinnedBuffer (truly asynchronous) or a C-contiguous NumPy array (the call returns when the runtime has staged
        it; the copy still runs on the copy stream, concurrently with the kernels of the main stream)"""
        assert dst.nbytes == src.nbytes
        ptr = src.ptr if isinstance(src, PinnedBuffer) else src.ctypes.data
        _check(self.lib.ssdseg_upload_async(self.handle, dst.ptr, ptr, dst.nbytes, 1 if after_fence else 0), "ssdseg_upload_async")

    def upload_fence(self):
        _check(self.lib.ssdseg_upload_fence(self.handle), "ssdseg_upload_fence")

    def upload_join(self):
        _check(self.lib.ssdseg_upload_join(self.handle), "ssdseg_upload_join")

    def upload_sync(self):
        _check(self.lib.ssdseg_upload_sync(self.handle), "ssdseg_upload_sync")

    # ---- generic call: ctx.call("ssdseg_pwconv_fwd", view, ldx, w, y, ...) with DeviceBuffer/None/int/float args
    def call(self, name: str, *args):
        fn = getattr(self.lib, name)
        conv = []
        for a in args:
            if isinstance(a, DeviceBuffer):
                conv.append(C.c_void_p(a.ptr))
            elif isinstance(a, (ViewStruct, GViewStruct)):
                conv.append(C.byref(a))
            else:
                conv.append(a)
        _check(fn(self.handle, *conv), name)

    def parts(self, name: str, *dims) -> int:
        out = C.c_int()
        _check(getattr(self.lib, name)(*[int(d) for d in dims], C.byref(out)), name)
        return out.value


def device_count() -> int:
    n = C.c_int()
    _check(load_library().ssdseg_device_count(C.byref(n)), "ssdseg_device_count")
    return n.value


def same_pad(size: int, k: int, s: int, d: int) -> Tuple[int, int, int]:
    """TF SAME geometry -> (out, pad_before, pad_after)  (SURVEY.md App. B.1)."""
    out = -(-size // s)
    keff = (k - 1) * d + 1
    total = max((out - 1) * s + keff - size, 0)
    return out, total // 2, total - total // 2
