"""Python handle of the HIP engine: device arrays and thin wrappers over the C ABI.

Host language note: the reference is 100 % Python, so the host side above the C ABI is Python and
mirrors the reference's call signatures (see utils_model.py / utils_sisr.py in this package).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib


class EngineError(RuntimeError):
    pass


class EngineRangeError(EngineError):
    """DPIR_ERR_RANGE: in f16x3 mode an activation left the f16 operand range and was clamped -- the images are wrong.
    Reload the model on an engine with set_precision('f32') (YAML: engine_precision: f32)."""


def _ptr(a) -> Optional[int]:
    """Raw device address of a DeviceArray / torch cuda tensor / int / None."""
    if a is None:
        return None
    if isinstance(a, DeviceArray):
        return a.ptr
    if isinstance(a, int):
        return a
    if hasattr(a, "data_ptr"):        # torch.Tensor (plumbing only)
        if not a.is_cuda or not a.is_contiguous():
            raise EngineError("torch tensors passed to the engine must be contiguous CUDA/HIP tensors")
        return a.data_ptr()
    raise TypeError(f"cannot take a device pointer of {type(a)}")


class DeviceArray:
    """A contiguous device buffer owned by an Engine (dpir_malloc)."""

    def __init__(self, engine: "Engine", shape: Sequence[int], dtype=np.float32):
        self.engine = engine
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        engine._check(engine.lib.dpir_malloc(engine.h, max(self.nbytes, 1), C.byref(p)))
        self.ptr = p.value

    def __del__(self):
        try:
            if self.ptr and self.engine.h:
                self.engine.lib.dpir_free(self.engine.h, self.ptr)
        except Exception:
            pass
        self.ptr = None

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def copy_from(self, host) -> "DeviceArray":
        a = np.ascontiguousarray(host, dtype=self.dtype)
        if a.shape != self.shape:
            raise EngineError(f"shape mismatch: {a.shape} vs {self.shape}")
        self.engine._check(self.engine.lib.dpir_h2d(self.engine.h, self.ptr, a.ctypes.data, self.nbytes))
        return self

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, self.dtype)
        self.engine._check(self.engine.lib.dpir_d2h(self.engine.h, out.ctypes.data, self.ptr, self.nbytes))
        return out

    # ---- the loop body's own arithmetic (main_ddpir.py:437, 440, 444), one rounded float32 operation per element like a torch
    # elementwise op (dpir_ewise); operands: DeviceArray of the same shape, a one-element DeviceArray (a 0-dim tensor such as
    # `norm`), or a Python / numpy scalar (cast to float32 as torch does for a float32 tensor)
    __array_ufunc__ = None          # numpy scalars (schedule-table entries) defer to __rmul__ / __radd__ ... below

    def _ew(self, op: int, other) -> "DeviceArray":
        if self.dtype != np.float32:
            raise EngineError("arithmetic is defined for float32 device arrays")
        out = DeviceArray(self.engine, self.shape, np.float32)
        e = self.engine
        if isinstance(other, DeviceArray):
            if other.size not in (1, self.size):
                raise EngineError(f"shape mismatch: {self.shape} vs {other.shape}")
            e._check(e.lib.dpir_ewise(e.h, op, self.ptr, other.ptr, other.size, 0.0, out.ptr, self.size))
        else:
            e._check(e.lib.dpir_ewise(e.h, op, self.ptr, None, 0, float(np.float32(other)), out.ptr, self.size))
        return out

    def __add__(self, o): return self._ew(0, o)
    def __radd__(self, o): return self._ew(0, o)
    def __sub__(self, o): return self._ew(1, o)
    def __rsub__(self, o): return self._ew(4, o)
    def __mul__(self, o): return self._ew(2, o)
    def __rmul__(self, o): return self._ew(2, o)
    def __truediv__(self, o): return self._ew(3, o)
    def __rtruediv__(self, o): return self._ew(5, o)
    def __neg__(self): return self._ew(2, -1.0)

    # torch idioms of the reference loop body that have no meaning for an engine array (gradients are the engine's tape)
    def requires_grad_(self, flag=True): return self
    def detach_(self): return self
    def detach(self): return self

    def clone(self) -> "DeviceArray":
        o = DeviceArray(self.engine, self.shape, self.dtype)
        self.engine._check(self.engine.lib.dpir_d2d(self.engine.h, o.ptr, self.ptr, self.nbytes))
        return o


class Engine:
    """One engine per GPU.  Raises EngineError / EngineLibraryError if the HIP path is unavailable."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.dpir_create(int(device), C.byref(h))
        if rc != 0 or not h.value:
            raise EngineError(f"dpir_create(device={device}) failed with status {rc}: no usable gfx950 GPU "
                              f"(the engine has no CPU fallback)")
        self.h = h
        self.device = device
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.dpir_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            msg = self.lib.dpir_last_error(self.h)
            cls = EngineRangeError if rc == -6 else EngineError
            raise cls(f"engine error {rc}: {msg.decode() if msg else ''}")

    # ---- memory
    def empty(self, shape, dtype=np.float32) -> DeviceArray:
        return DeviceArray(self, shape, dtype)

    def to_device(self, host, dtype=None) -> DeviceArray:
        a = np.ascontiguousarray(host if dtype is None else np.asarray(host, dtype=dtype))
        return DeviceArray(self, a.shape, a.dtype).copy_from(a)

    def sync(self):
        self._check(self.lib.dpir_sync(self.h))

    # ---- UNet
    def set_precision(self, mode):
        """'f32' (exact fp32 MFMA), 'f16x3' (operand-split f16 MFMA, fp32-equivalent accuracy) or 'f16x1' (f16 operands, fp32
        accumulation: the reference's use_fp16 recipe -- a REDUCED-precision mode with its own quality contract); before load_unet."""
        m = {"f32": 0, "f16x3": 1, "f16x1": 2, 0: 0, 1: 1, 2: 2}[mode]
        self._check(self.lib.dpir_set_precision(self.h, m))
        self.precision = ("f32", "f16x3", "f16x1")[m]

    precision = "f32"

    def set_prox_launch(self, mode):
        """'wave' / 1: wave-per-transform kernels on a column-major spectrum (csrc/fft4.hip, 256 x 256 and 512 x 512; default); 'launches' / 0: the two-pass register
        kernels (csrc/fft2.hip).  Takes effect at the next pre_calculate / loop call."""
        m = {"wave": 1, "launches": 0}.get(mode, mode)
        self._check(self.lib.dpir_set_prox_launch(self.h, int(m)))

    def enable_grad(self, on=True):
        """Gradient mode (generate_mode 'DPS_y0'): before load_unet / load_state_dict."""
        self._check(self.lib.dpir_enable_grad(self.h, 1 if on else 0))
        self.grad = bool(on)

    grad = False

    def unet_vjp(self, x, t, gout, y=None, out=None, dx=None):
        """(forward output, J(x)^T gout): the input-gradient of the UNet that torch.autograd.grad computes in the reference."""
        B, _, H, W = x.shape
        t = np.ascontiguousarray(t, dtype=np.int64)
        yv = None if y is None else np.ascontiguousarray(y, dtype=np.int64)
        if out is None:
            out = self.empty((B, self.out_channels, H, W))
        if dx is None:
            dx = self.empty((B, 3, H, W))
        self._check(self.lib.dpir_unet_vjp(self.h, _ptr(x), t.ctypes.data, None if yv is None else yv.ctypes.data, _ptr(gout),
                                           _ptr(out), _ptr(dx), B, H, W))
        return out, dx

    def p_sample(self, x, t: int, coef: "_lib.PSampleCoef", noise, y=None, xt=None, x0=None):
        """(sample, pred_xstart) of GaussianDiffusion.p_sample / ddim_sample(eta=0) around one denoiser call (dpir_p_sample)."""
        B, _, H, W = x.shape
        yv = None if y is None else np.ascontiguousarray(y, dtype=np.int64)
        xt = self.empty((B, 3, H, W)) if xt is None else xt
        x0 = self.empty((B, 3, H, W)) if x0 is None else x0
        self._check(self.lib.dpir_p_sample(self.h, _ptr(x), int(t), C.byref(coef), _ptr(noise), None if yv is None else yv.ctypes.data,
                                           _ptr(xt), _ptr(x0), B, H, W))
        return xt, x0

    def grad_and_value(self, through_network: bool, x_hat, measurement, sf: int):
        """(norm_grad [B,3,H,W], norm [1]) of || measurement - Resizer_{1/sf}(x_hat) ||_2 (dpir_grad_and_value)."""
        B, _, H, W = x_hat.shape
        g, nv = self.empty((B, 3, H, W)), self.empty((1,))
        self._check(self.lib.dpir_grad_and_value(self.h, 1 if through_network else 0, _ptr(x_hat), _ptr(measurement), int(sf), g.ptr, nv.ptr, B, H, W))
        return g, nv

    def load_unet(self, desc: "_lib.UNetDesc", state_dict: Dict[str, np.ndarray]):
        n = len(state_dict)
        arr = (_lib.Tensor * n)()
        keep = []
        for i, (k, v) in enumerate(state_dict.items()):
            if hasattr(v, "detach"):           # torch tensor from torch.load (checkpoint loading only)
                v = v.detach().cpu().numpy()
            a = np.ascontiguousarray(v, dtype=np.float32)
            keep.append(a)
            arr[i].name = k.encode()
            arr[i].data = a.ctypes.data
            arr[i].ndim = a.ndim
            for d in range(a.ndim):
                arr[i].shape[d] = a.shape[d]
        self._check(self.lib.dpir_load_unet(self.h, C.byref(desc), arr, n))

    def unet_forward(self, x, t, y=None, out=None):
        B, _, H, W = x.shape
        t = np.ascontiguousarray(t, dtype=np.int64)
        yv = None if y is None else np.ascontiguousarray(y, dtype=np.int64)
        oc = self.out_channels
        if out is None:
            out = self.empty((B, oc, H, W))
        self._check(self.lib.dpir_unet_forward(self.h, _ptr(x), t.ctypes.data, None if yv is None else yv.ctypes.data,
                                               _ptr(out), B, H, W))
        return out

    out_channels = 6

    def model_fn_xstart(self, x, t: int, c1: float, c2: float, y=None, out=None):
        B, _, H, W = x.shape
        yv = None if y is None else np.ascontiguousarray(y, dtype=np.int64)
        if out is None:
            out = self.empty((B, 3, H, W))
        self._check(self.lib.dpir_model_fn_xstart(self.h, _ptr(x), int(t), float(c1), float(c2),
                                                  None if yv is None else yv.ctypes.data, _ptr(out), B, H, W))
        return out

    def read_tap(self, name: str) -> np.ndarray:
        n = C.c_size_t()
        self._check(self.lib.dpir_unet_read_tap(self.h, name.encode(), None, 0, C.byref(n)))
        out = np.empty(n.value, np.float32)
        self._check(self.lib.dpir_unet_read_tap(self.h, name.encode(), out.ctypes.data, n.value, C.byref(n)))
        return out

    def unet_flops(self, H, W, cls: int = -1) -> float:
        return float(self.lib.dpir_unet_flops_class(self.h, H, W, cls))

    def graph_cache_size(self) -> int:
        return int(self.lib.dpir_graph_cache_size(self.h))

    # ---- profiling
    def prof_enable(self, on=True):
        self._check(self.lib.dpir_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self._check(self.lib.dpir_prof_reset(self.h))

    def prof_read(self):
        ms = (C.c_double * _lib.PROF_CLASSES)()
        cnt = (C.c_int64 * _lib.PROF_CLASSES)()
        self._check(self.lib.dpir_prof_read(self.h, ms, cnt))
        return {_lib.PROF_NAMES[i]: (ms[i], cnt[i]) for i in range(_lib.PROF_CLASSES)}


def device_count() -> int:
    """HIP devices visible to this process (dpir_device_count; the reference: torch.cuda.device_count(), main_ddpir.py:135)."""
    n = C.c_int(0)
    _lib.load().dpir_device_count(C.byref(n))
    return int(n.value)


_default: Dict[int, Engine] = {}


def default_engine(device: int = 0) -> Engine:
    if device not in _default:
        _default[device] = Engine(device)
    return _default[device]
