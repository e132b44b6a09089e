"""MI355X-native LceBconv2d / LceQuantize hot path -- thin Python binding of the C ABI.

The product is the C-ABI shared library ``csrc/liblce_hip.so`` (hand-written gfx950
kernels, declared in ``include/lce_hip.h``) plus the C++ TFLite op glue in
``csrc/tflite/``.  This module only loads the library with ``ctypes`` and passes raw
device pointers to it (PyTorch is used by callers purely to own HBM buffers and
streams).  There is no fallback of any kind: if the library is missing or there is no
GPU, calls raise.

The directory name contains a hyphen, so import it with
``importlib.import_module("compute-engine_amd")``.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LCE_HIP_LIBRARY: an alternative build of the same library (kernel A/B experiments, tools/)
LIB_PATH = os.environ.get("LCE_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "liblce_hip.so")

# enums of include/lce_hip.h
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_RUNTIME, ERR_NO_DEVICE = range(5)
F32, I8, BITPACKED, BOOL = range(4)
PADDING_SAME, PADDING_VALID = 0, 1
ACT_NONE, ACT_RELU, ACT_RELU_N1_TO_1, ACT_RELU6 = range(4)
SEM_REFERENCE, SEM_OPTIMIZED = 0, 1

# every symbol include/lce_hip.h declares (tests check the library exports them all)
ABI_SYMBOLS = (
    "lce_hip_abi_version", "lce_hip_last_error", "lce_hip_build_flavor", "lce_hip_device_count", "lce_hip_set_device",
    "lce_hip_malloc", "lce_hip_free", "lce_hip_memcpy_h2d", "lce_hip_memcpy_d2h", "lce_hip_memset",
    "lce_hip_host_register", "lce_hip_host_unregister",
    "lce_hip_stream_create", "lce_hip_stream_destroy", "lce_hip_stream_synchronize",
    "lce_hip_graph_begin_capture", "lce_hip_graph_end_capture", "lce_hip_graph_launch", "lce_hip_graph_destroy",
    "lce_hip_bitpacked_size", "lce_hip_bitpack", "lce_hip_unpack",
    "lce_hip_bconv2d_plan_create", "lce_hip_bconv2d_plan_destroy", "lce_hip_bconv2d_plan_output_shape",
    "lce_hip_bconv2d_plan_padding", "lce_hip_bconv2d_plan_set_weights", "lce_hip_bconv2d_plan_folded",
    "lce_hip_bconv2d_plan_set_option", "lce_hip_bconv2d_plan_kernel_name", "lce_hip_bconv2d_plan_kernel_name_dual", "lce_hip_bconv2d_plan_int8_epilogue", "lce_hip_bconv2d_run",
    "lce_hip_bconv2d_run_dual", "lce_hip_bconv2d_plan_device", "lce_hip_bconv2d_run_host", "lce_hip_bmaxpool_output_shape", "lce_hip_bmaxpool",
    "lce_hip_prepare_binary_filter", "lce_hip_prepare_fuse_post_op", "lce_hip_prepare_can_fuse_activation",
    "lce_hip_prepare_bitpacked_output", "lce_hip_prepare_bitpack_filter",
)
POST_ADD, POST_SUB, POST_MUL, POST_DIV = 0, 1, 2, 3


class LceHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"lce_hip error {code}: {message}")
        self.code = code
        self.message = message


class Bconv2dDesc(C.Structure):
    """``lce_hip_bconv2d_desc``."""
    _fields_ = [(n, C.c_int32) for n in (
        "batch", "in_height", "in_width", "channels_in", "filter_height", "filter_width",
        "channels_out", "groups", "stride_height", "stride_width", "dilation_height",
        "dilation_width", "padding", "pad_values", "activation", "dst_type", "semantics")] + [
        ("out_scale", C.c_float), ("out_zero_point", C.c_int32)]


_lib = None


def lib() -> C.CDLL:
    """Load ``liblce_hip.so``; raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `make -C compute-engine_amd/csrc` "
                "(there is no Python or CPU fallback for the HIP kernels)")
        l = C.CDLL(LIB_PATH)
        l.lce_hip_last_error.restype = C.c_char_p
        l.lce_hip_build_flavor.restype = C.c_char_p
        l.lce_hip_bconv2d_plan_kernel_name.restype = C.c_char_p
        l.lce_hip_bconv2d_plan_kernel_name.argtypes = [C.c_void_p]
        l.lce_hip_bconv2d_plan_kernel_name_dual.restype = C.c_char_p
        l.lce_hip_bconv2d_plan_kernel_name_dual.argtypes = [C.c_void_p]
        l.lce_hip_bconv2d_plan_int8_epilogue.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        l.lce_hip_bconv2d_plan_destroy.restype = None
        l.lce_hip_bconv2d_plan_destroy.argtypes = [C.c_void_p]
        l.lce_hip_bconv2d_plan_create.argtypes = [C.POINTER(Bconv2dDesc), C.POINTER(C.c_void_p)]
        l.lce_hip_bconv2d_plan_output_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        l.lce_hip_bconv2d_plan_padding.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        l.lce_hip_bconv2d_plan_set_weights.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        l.lce_hip_bconv2d_plan_folded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        l.lce_hip_bconv2d_plan_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        l.lce_hip_bconv2d_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.lce_hip_bconv2d_run_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        l.lce_hip_bconv2d_run_dual.argtypes = [C.c_void_p] * 5
        l.lce_hip_host_register.argtypes = [C.c_void_p, C.c_size_t]
        l.lce_hip_host_unregister.argtypes = [C.c_void_p]
        l.lce_hip_bconv2d_plan_device.argtypes = [C.c_void_p]
        l.lce_hip_bitpack.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int32,
                                      C.c_void_p, C.c_void_p]
        l.lce_hip_unpack.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float,
                                     C.c_int32, C.c_void_p, C.c_void_p]
        l.lce_hip_bmaxpool.argtypes = [C.c_void_p] + [C.c_int32] * 9 + [C.c_void_p, C.c_void_p]
        l.lce_hip_bmaxpool_output_shape.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_int32)] * 2
        _lib = l
    return _lib


def check(code: int) -> None:
    if code != OK:
        raise LceHipError(code, lib().lce_hip_last_error().decode(errors="replace"))


def device_count() -> int:
    return lib().lce_hip_device_count()


def bitpacked_size(n: int) -> int:
    return (n + 31) // 32


def _host_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class ConvParams:
    """The LceBconv2d attributes (tflite/kernels/bconv2d.cc:94-124) plus tensor metadata."""
    batch: int
    in_height: int
    in_width: int
    channels_in: int
    filter_height: int
    filter_width: int
    channels_out: int
    groups: int = 1
    stride_height: int = 1
    stride_width: int = 1
    dilation_height: int = 1
    dilation_width: int = 1
    padding: int = PADDING_VALID
    pad_values: int = 0
    activation: int = ACT_NONE
    dst_type: int = F32
    semantics: int = SEM_OPTIMIZED
    out_scale: float = 1.0
    out_zero_point: int = 0

    def desc(self) -> Bconv2dDesc:
        return Bconv2dDesc(self.batch, self.in_height, self.in_width, self.channels_in,
                           self.filter_height, self.filter_width, self.channels_out, self.groups,
                           self.stride_height, self.stride_width, self.dilation_height,
                           self.dilation_width, self.padding, self.pad_values, self.activation,
                           self.dst_type, self.semantics, float(self.out_scale),
                           int(self.out_zero_point))


class Bconv2dPlan:
    """Owns one ``lce_hip_bconv2d_plan`` (Prepare + OneTimeSetup of one LceBconv2d node)."""

    def __init__(self, params: ConvParams):
        self.params = params
        self._h = C.c_void_p()
        d = params.desc()
        check(lib().lce_hip_bconv2d_plan_create(C.byref(d), C.byref(self._h)))
        dims = (C.c_int32 * 4)()
        check(lib().lce_hip_bconv2d_plan_output_shape(self._h, dims))
        self.output_shape = tuple(dims)

    @classmethod
    def from_handle(cls, handle: int, dst_type: int, channels_out: int = 0) -> "Bconv2dPlan":
        """Adopts a plan created through the C ABI elsewhere (lce_tflite_model_bconv2d_plan)."""
        import types
        self = cls.__new__(cls)
        self.params = types.SimpleNamespace(dst_type=dst_type, channels_out=channels_out)
        self._h = C.c_void_p(handle)
        dims = (C.c_int32 * 4)()
        check(lib().lce_hip_bconv2d_plan_output_shape(self._h, dims))
        self.output_shape = tuple(dims)
        return self

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value and _lib is not None:
            _lib.lce_hip_bconv2d_plan_destroy(self._h)   # (_lib is None again during interpreter teardown)
            self._h = C.c_void_p()

    __del__ = close

    def padding(self):
        ph, pw = C.c_int32(), C.c_int32()
        check(lib().lce_hip_bconv2d_plan_padding(self._h, C.byref(ph), C.byref(pw)))
        return ph.value, pw.value

    def set_weights(self, filter_ohwi, post_mul=None, post_bias=None, thresholds=None):
        """Host numpy arrays: int32 OHWI filter words, float32 [Cout] x2, int32 [Cout]."""
        import numpy as np
        f = np.ascontiguousarray(filter_ohwi, np.int32)
        m = None if post_mul is None else np.ascontiguousarray(post_mul, np.float32)
        b = None if post_bias is None else np.ascontiguousarray(post_bias, np.float32)
        t = None if thresholds is None else np.ascontiguousarray(thresholds, np.int32)
        p = self.params
        if hasattr(p, "filter_height"):   # (a plan adopted from a handle carries no descriptor: the C side owns it)
            # the C ABI reads exactly this many elements from the raw pointers
            want = p.channels_out * p.filter_height * p.filter_width * bitpacked_size(p.channels_in // max(1, p.groups))
            if f.size != want:
                raise ValueError(f"filter has {f.size} words, the plan needs Cout*KH*KW*ceil(Cin/G/32) = {want}")
            for name, a in (("post_activation_multiplier", m), ("post_activation_bias", b), ("thresholds", t)):
                if a is not None and a.size != p.channels_out:
                    raise ValueError(f"{name} has {a.size} entries, the plan needs channels_out = {p.channels_out}")
        check(lib().lce_hip_bconv2d_plan_set_weights(self._h, _host_ptr(f), _host_ptr(m),
                                                     _host_ptr(b), _host_ptr(t)))

    def folded(self):
        import numpy as np
        n = self.params.channels_out
        mul, bias = np.empty(n, np.float32), np.empty(n, np.float32)
        lo, hi = C.c_int32(), C.c_int32()
        check(lib().lce_hip_bconv2d_plan_folded(self._h, _host_ptr(mul), _host_ptr(bias),
                                                C.byref(lo), C.byref(hi)))
        return mul, bias, lo.value, hi.value

    def set_option(self, key: str, value: str):
        check(lib().lce_hip_bconv2d_plan_set_option(self._h, key.encode(), value.encode()))

    def kernel_name(self, dual: bool = False) -> str:
        if dual:
            return lib().lce_hip_bconv2d_plan_kernel_name_dual(self._h).decode()
        return lib().lce_hip_bconv2d_plan_kernel_name(self._h).decode()

    def int8_epilogue(self):
        """(one_instruction_forms, adjusted_channels) of the kernel the next run launches: ``lce_hip_bconv2d_plan_int8_epilogue``."""
        forms, adjusted = C.c_int32(), C.c_int32()
        check(lib().lce_hip_bconv2d_plan_int8_epilogue(self._h, C.byref(forms), C.byref(adjusted)))
        return bool(forms.value), int(adjusted.value)

    def run_ptr(self, input_dev: int, output_dev: int, stream: int = 0):
        check(lib().lce_hip_bconv2d_run(self._h, C.c_void_p(input_dev), C.c_void_p(output_dev),
                                        C.c_void_p(stream)))

    def device(self) -> int:
        """The HIP device the plan's buffers live on (-1 before the first run)."""
        return lib().lce_hip_bconv2d_plan_device(self._h)

    def _check_input(self, x):
        import torch
        assert x.is_cuda and x.dtype == torch.int32 and x.is_contiguous()
        p = self.params
        if hasattr(p, "in_height"):
            want = (p.batch, p.in_height, p.in_width, bitpacked_size(p.channels_in))
            if tuple(x.shape) != want:
                raise ValueError(f"input shape {tuple(x.shape)} does not match the plan's {want}")

    def _out(self, x, out, dtype):
        import torch
        if out is None:
            return torch.empty(self.output_shape, dtype=dtype, device=x.device)
        assert out.is_cuda and out.device == x.device and out.dtype == dtype and out.is_contiguous()
        if tuple(out.shape) != tuple(self.output_shape):
            raise ValueError(f"output shape {tuple(out.shape)} does not match the plan's {tuple(self.output_shape)}")
        return out

    def run(self, x, out=None, stream: int | None = None):
        """x: CUDA int32 tensor [B,H,W,ceil(Cin/32)]; returns / fills the output tensor."""
        import torch
        self._check_input(x)
        out = self._out(x, out, {F32: torch.float32, I8: torch.int8, BITPACKED: torch.int32}[self.params.dst_type])
        with torch.cuda.device(x.device):   # the plan's buffers land on (and must stay on) the tensor's device
            if stream is None:
                stream = torch.cuda.current_stream(x.device).cuda_stream
            self.run_ptr(x.data_ptr(), out.data_ptr(), stream)
        return out

    def run_dual(self, x, out=None, out_bits=None, stream: int | None = None):
        """Float or int8 output AND its LceQuantize ([B,OH,OW,ceil(Cout/32)] int32: sign bits, or for an int8 plan
        bit = q < out_zero_point) in one pass."""
        import torch
        self._check_input(x)
        if self.params.dst_type == BITPACKED:
            raise ValueError("run_dual: the plan already writes bits")
        out = self._out(x, out, torch.float32 if self.params.dst_type == F32 else torch.int8)
        b, oh, ow, n = self.output_shape
        if out_bits is None:
            out_bits = torch.empty((b, oh, ow, bitpacked_size(n)), dtype=torch.int32, device=x.device)
        assert out_bits.is_cuda and out_bits.dtype == torch.int32 and out_bits.is_contiguous()
        if tuple(out_bits.shape) != (b, oh, ow, bitpacked_size(n)):
            raise ValueError(f"bit output shape {tuple(out_bits.shape)} does not match {(b, oh, ow, bitpacked_size(n))}")
        with torch.cuda.device(x.device):
            if stream is None:
                stream = torch.cuda.current_stream(x.device).cuda_stream
            check(lib().lce_hip_bconv2d_run_dual(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(out_bits.data_ptr()), C.c_void_p(stream)))
        return out, out_bits

    def run_host(self, x_np, out=None):
        """Host (NumPy) tensors through H2D | kernel | D2H; pass page-locked arrays (``host_register``) and an
        ``out`` to reuse for fully overlapped copies."""
        import numpy as np
        x = np.ascontiguousarray(x_np, np.int32)
        dt = {F32: np.float32, I8: np.int8, BITPACKED: np.int32}[self.params.dst_type]
        if out is None:
            out = np.empty(self.output_shape, dt)
        assert out.dtype == dt and out.flags.c_contiguous and tuple(out.shape) == tuple(self.output_shape)
        check(lib().lce_hip_bconv2d_run_host(self._h, _host_ptr(x), _host_ptr(out)))
        return out


class host_register:
    """``with host_register(a, b): ...`` page-locks the NumPy arrays for the duration of the block."""

    def __init__(self, *arrays):
        self.arrays = arrays

    def __enter__(self):
        self.done = []
        for a in self.arrays:
            check(lib().lce_hip_host_register(_host_ptr(a), C.c_size_t(a.nbytes)))
            self.done.append(a)
        return self

    def __exit__(self, *exc):
        for a in self.done:
            lib().lce_hip_host_unregister(_host_ptr(a))
        return False


def bitpack(x, zero_point: int = 0, out=None, stream: int | None = None):
    """LceQuantize on a CUDA tensor (float32 / int8 / bool), packing the last axis."""
    import torch
    assert x.is_cuda and x.is_contiguous()
    t = {torch.float32: F32, torch.int8: I8, torch.bool: BOOL}[x.dtype]
    cols = x.shape[-1]
    rows = x.numel() // cols if cols else 0
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (bitpacked_size(cols),), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        if stream is None:
            stream = torch.cuda.current_stream(x.device).cuda_stream
        check(lib().lce_hip_bitpack(t, C.c_void_p(x.data_ptr()), rows, cols, int(zero_point),
                                    C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
    return out


def unpack(words, channels: int, dtype, scale: float = 1.0, zero_point: int = 0, stream: int | None = None, out=None):
    """LceDequantize on a CUDA int32 tensor."""
    import torch
    assert words.is_cuda and words.dtype == torch.int32 and words.is_contiguous()
    t = {torch.float32: F32, torch.int8: I8, torch.bool: BOOL}[dtype]
    rows = words.numel() // words.shape[-1]
    if out is None:
        out = torch.empty(tuple(words.shape[:-1]) + (channels,), dtype=dtype, device=words.device)
    assert out.is_cuda and out.dtype == dtype and out.is_contiguous() and tuple(out.shape) == tuple(words.shape[:-1]) + (channels,)
    with torch.cuda.device(words.device):
        if stream is None:
            stream = torch.cuda.current_stream(words.device).cuda_stream
        check(lib().lce_hip_unpack(t, C.c_void_p(words.data_ptr()), rows, channels, float(scale),
                                   int(zero_point), C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
    return out


def bmaxpool(x, filter_height, filter_width, stride_height, stride_width, padding, stream: int | None = None, out=None):
    """LceBMaxPool2d on a CUDA int32 tensor [B,H,W,words]."""
    import torch
    assert x.is_cuda and x.dtype == torch.int32 and x.is_contiguous() and x.dim() == 4
    b, h, w, c = x.shape
    oh, ow = C.c_int32(), C.c_int32()
    check(lib().lce_hip_bmaxpool_output_shape(h, w, filter_height, filter_width, stride_height,
                                              stride_width, padding, C.byref(oh), C.byref(ow)))
    if out is None:
        out = torch.empty((b, oh.value, ow.value, c), dtype=torch.int32, device=x.device)
    assert out.is_cuda and out.dtype == torch.int32 and out.is_contiguous() and tuple(out.shape) == (b, oh.value, ow.value, c)
    with torch.cuda.device(x.device):
        if stream is None:
            stream = torch.cuda.current_stream(x.device).cuda_stream
        check(lib().lce_hip_bmaxpool(C.c_void_p(x.data_ptr()), b, h, w, c, filter_height, filter_width,
                                     stride_height, stride_width, padding, C.c_void_p(out.data_ptr()),
                                     C.c_void_p(stream)))
    return out


# ---------------------------------------------------------------------------------------
# converter-side parameter preparation (host-only; include/lce_hip.h "lce_hip_prepare_*")
# ---------------------------------------------------------------------------------------
def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def prepare_binary_filter(filter_hwio):
    """float HWIO +-scale filter -> (OHWI +-1 filter, post_activation_multiplier, post_activation_bias)."""
    f = _f32(filter_hwio)
    kh, kw, cin, cout = f.shape
    ohwi = np.empty((cout, kh, kw, cin), np.float32)
    mul, bias = np.empty(cout, np.float32), np.empty(cout, np.float32)
    check(lib().lce_hip_prepare_binary_filter(_host_ptr(f), kh, kw, cin, cout, _host_ptr(ohwi),
                                              _host_ptr(mul), _host_ptr(bias)))
    return ohwi, mul, bias


def prepare_fuse_post_op(op: int, value, mul, bias):
    """Fuses ``conv <op> value`` into (mul, bias); returns new arrays."""
    v = _f32(np.atleast_1d(value))
    m, b = _f32(mul).copy(), _f32(bias).copy()
    check(lib().lce_hip_prepare_fuse_post_op(op, _host_ptr(v), v.size, _host_ptr(m), _host_ptr(b), m.size))
    return m, b


def prepare_can_fuse_activation(mul, bias, padding: int, pad_values: int) -> bool:
    m, b = _f32(mul), _f32(bias)
    return bool(lib().lce_hip_prepare_can_fuse_activation(_host_ptr(m), _host_ptr(b), m.size, padding, pad_values))


def prepare_bitpacked_output(filter_ohwi, mul, bias, activation: int = ACT_NONE,
                             padding: int = PADDING_VALID, pad_values: int = 0):
    """-> (sign-flipped OHWI filter, int32 thresholds) of the bit-writing convolution."""
    f = _f32(filter_ohwi).copy()
    cout, kh, kw, cin = f.shape
    m, b = _f32(mul), _f32(bias)
    thr = np.empty(cout, np.int32)
    check(lib().lce_hip_prepare_bitpacked_output(_host_ptr(f), kh, kw, cin, cout, activation, padding, pad_values,
                                                 _host_ptr(m), _host_ptr(b), _host_ptr(thr)))
    return f, thr


def prepare_bitpack_filter(filter_ohwi):
    """float OHWI filter -> int32 [O, H, W, ceil(I/32)] (input 1 of LceBconv2d)."""
    f = _f32(filter_ohwi)
    cout, kh, kw, cin = f.shape
    words = np.empty((cout, kh, kw, (cin + 31) // 32), np.int32)
    check(lib().lce_hip_prepare_bitpack_filter(_host_ptr(f), kh, kw, cin, cout, _host_ptr(words)))
    return words
