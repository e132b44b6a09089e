"""ctypes binding of oracle/liblce_oracle.so -- the CPU restatement of the reference.

Test infrastructure only (see oracle/lce_oracle.h).  Imported by tests/,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``; never by
the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")

PADDING_SAME, PADDING_VALID = 0, 1
ACT_NONE, ACT_RELU, ACT_RELU_N1_TO_1, ACT_RELU6 = 0, 1, 2, 3
DST_F32, DST_I8, DST_BITPACKED = 0, 1, 2
SEM_REFERENCE, SEM_OPTIMIZED = 0, 1


class _Conv(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "batch", "in_h", "in_w", "channels_in", "filter_h", "filter_w", "channels_out",
        "groups", "stride_h", "stride_w", "dilation_h", "dilation_w", "padding",
        "pad_values", "activation", "semantics", "out_h", "out_w", "pad_h", "pad_w",
        "pad_h_offset", "pad_w_offset")]


def _build_if_needed(path: str) -> None:
    src = os.path.join(_ORACLE_DIR, "lce_oracle.c")
    if os.path.exists(path) and os.path.getmtime(path) >= os.path.getmtime(src):
        return
    subprocess.run(["make", "-C", _ORACLE_DIR], check=True, capture_output=True)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(_ORACLE_DIR, "liblce_oracle.so")
        _build_if_needed(path)
        _lib = C.CDLL(path)
        _lib.lce_oracle_zero_pad_cache_size.restype = C.c_size_t
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def bitpacked_size(n: int) -> int:
    return (n + 31) // 32


@dataclass
class ConvSpec:
    """Mirror of BConv2DParams + the op attributes (core/bconv2d/params.h:12-32,
    tflite/kernels/bconv2d.cc:94-124)."""
    batch: int
    in_h: int
    in_w: int
    channels_in: int
    filter_h: int
    filter_w: int
    channels_out: int
    groups: int = 1
    stride_h: int = 1
    stride_w: int = 1
    dilation_h: int = 1
    dilation_w: int = 1
    padding: int = PADDING_VALID
    pad_values: int = 0
    activation: int = ACT_NONE
    semantics: int = SEM_REFERENCE
    _c: _Conv = field(default=None, repr=False, compare=False)

    def with_batch(self, batch: int) -> "ConvSpec":
        """Same layer at another batch size (dataclasses.replace would carry the cached C struct)."""
        import dataclasses
        return dataclasses.replace(self, batch=batch, _c=None)

    def c_struct(self) -> _Conv:
        if self._c is None:
            c = _Conv()
            for n in ("batch", "in_h", "in_w", "channels_in", "filter_h", "filter_w",
                      "channels_out", "groups", "stride_h", "stride_w", "dilation_h",
                      "dilation_w", "padding", "pad_values", "activation", "semantics"):
                setattr(c, n, getattr(self, n))
            rc = lib().lce_oracle_conv_prepare(C.byref(c))
            if rc != 0:
                raise ValueError(f"invalid conv spec (code {rc}): {self}")
            self._c = c
        return self._c

    @property
    def out_h(self): return self.c_struct().out_h
    @property
    def out_w(self): return self.c_struct().out_w
    @property
    def pad_h(self): return self.c_struct().pad_h
    @property
    def pad_w(self): return self.c_struct().pad_w
    @property
    def in_words(self): return bitpacked_size(self.channels_in)
    @property
    def filter_words(self): return bitpacked_size(self.channels_in // self.groups)
    @property
    def out_words(self): return bitpacked_size(self.channels_out)
    @property
    def backtransform_add(self): return self.filter_h * self.filter_w * (self.channels_in // self.groups)
    @property
    def binary_macs(self):
        return (self.batch * self.out_h * self.out_w * self.channels_out * self.filter_h *
                self.filter_w * (self.channels_in // self.groups))

    def input_shape(self): return (self.batch, self.in_h, self.in_w, self.in_words)
    def filter_shape(self): return (self.channels_out, self.filter_h, self.filter_w, self.filter_words)

    def output_shape(self, dst_type: int):
        last = self.out_words if dst_type == DST_BITPACKED else self.channels_out
        return (self.batch, self.out_h, self.out_w, last)


# --------------------------------------------------------------------------- bitpacking

def bitpack(x: np.ndarray, zero_point: int = 0) -> np.ndarray:
    """LceQuantize: pack along the last axis (bitpacking/utils.h:23-33)."""
    cols = x.shape[-1]
    rows = int(np.prod(x.shape[:-1], dtype=np.int64)) if x.ndim > 1 else 1
    x = np.ascontiguousarray(x)
    out = np.empty(x.shape[:-1] + (bitpacked_size(cols),), dtype=np.int32)
    if x.dtype == np.float32:
        assert zero_point == 0
        lib().lce_oracle_bitpack_f32(_p(x), C.c_size_t(rows), C.c_size_t(cols), _p(out))
    elif x.dtype == np.int8:
        lib().lce_oracle_bitpack_i8(_p(x), C.c_size_t(rows), C.c_size_t(cols),
                                    C.c_int32(zero_point), _p(out))
    elif x.dtype in (np.bool_, np.uint8):
        xv = x.view(np.uint8)
        lib().lce_oracle_bitpack_bool(_p(xv), C.c_size_t(rows), C.c_size_t(cols), _p(out))
    else:
        raise TypeError(x.dtype)
    return out


def unpack(words: np.ndarray, cols: int, dtype, scale: float = 1.0, zero_point: int = 0) -> np.ndarray:
    """LceDequantize (tflite/kernels/quantization.cc:116-147)."""
    words = np.ascontiguousarray(words, dtype=np.int32)
    rows = int(np.prod(words.shape[:-1], dtype=np.int64)) if words.ndim > 1 else 1
    out = np.empty(words.shape[:-1] + (cols,), dtype=dtype)
    if dtype == np.float32:
        lib().lce_oracle_unpack_f32(_p(words), C.c_size_t(rows), C.c_size_t(cols), _p(out))
    elif dtype == np.int8:
        lib().lce_oracle_unpack_i8(_p(words), C.c_size_t(rows), C.c_size_t(cols),
                                   C.c_float(scale), C.c_int32(zero_point), _p(out))
    elif dtype == np.bool_:
        o8 = out.view(np.uint8)
        lib().lce_oracle_unpack_bool(_p(words), C.c_size_t(rows), C.c_size_t(cols), _p(o8))
    else:
        raise TypeError(dtype)
    return out


# --------------------------------------------------------------------------- bconv2d

def fold_output_transform(spec: ConvSpec, dst_type: int, post_mul, post_bias,
                          out_scale: float = 1.0, out_zero_point: int = 0):
    n = spec.channels_out
    post_mul = np.ascontiguousarray(post_mul, dtype=np.float32)
    post_bias = np.ascontiguousarray(post_bias, dtype=np.float32)
    mul = np.empty(n, np.float32)
    bias = np.empty(n, np.float32)
    cmin, cmax = C.c_int32(), C.c_int32()
    lib().lce_oracle_fold_output_transform(C.byref(spec.c_struct()), C.c_int(dst_type),
                                           _p(post_mul), _p(post_bias), C.c_float(out_scale),
                                           C.c_int32(out_zero_point), _p(mul), _p(bias),
                                           C.byref(cmin), C.byref(cmax))
    return mul, bias, cmin.value, cmax.value


def thresholds_converter(spec: ConvSpec, post_mul, post_bias) -> np.ndarray:
    post_mul = np.ascontiguousarray(post_mul, dtype=np.float32)
    post_bias = np.ascontiguousarray(post_bias, dtype=np.float32)
    thr = np.empty(len(post_mul), np.int32)
    lib().lce_oracle_thresholds_converter(C.c_int32(spec.backtransform_add),
                                          C.c_int(spec.activation), _p(post_mul), _p(post_bias),
                                          C.c_int(len(post_mul)), _p(thr))
    return thr


def thresholds_optest(spec: ConvSpec, post_mul, post_bias) -> np.ndarray:
    post_mul = np.ascontiguousarray(post_mul, dtype=np.float32)
    post_bias = np.ascontiguousarray(post_bias, dtype=np.float32)
    thr = np.empty(len(post_mul), np.int32)
    lib().lce_oracle_thresholds_optest(C.c_int32(spec.backtransform_add),
                                       C.c_int(spec.activation), _p(post_mul), _p(post_bias),
                                       C.c_int(len(post_mul)), _p(thr))
    return thr


def zero_pad_cache(spec: ConvSpec, filt: np.ndarray, post_mul) -> np.ndarray:
    post_mul = np.ascontiguousarray(post_mul, dtype=np.float32)
    size = lib().lce_oracle_zero_pad_cache_size(C.byref(spec.c_struct()))
    cache = np.empty(size, np.float32)
    lib().lce_oracle_zero_pad_cache_fill(C.byref(spec.c_struct()), _p(filt), _p(post_mul), _p(cache))
    return cache


def bconv2d_accum(spec: ConvSpec, inp: np.ndarray, filt: np.ndarray, threads: int = 1) -> np.ndarray:
    out = np.empty(spec.output_shape(DST_F32), np.int32)
    lib().lce_oracle_bconv2d_accum(C.byref(spec.c_struct()), _p(inp), _p(filt), _p(out), C.c_int(threads))
    return out


def bconv2d(spec: ConvSpec, dst_type: int, inp: np.ndarray, filt: np.ndarray,
            post_mul=None, post_bias=None, thresholds=None, out_scale: float = 1.0,
            out_zero_point: int = 0, threads: int = 1) -> np.ndarray:
    """Full op: OneTimeSetup folding + Eval, following the registration named by
    ``spec.semantics`` (tflite/kernels/bconv2d.cc:324-392,416-516)."""
    inp = np.ascontiguousarray(inp, dtype=np.int32)
    filt = np.ascontiguousarray(filt, dtype=np.int32)
    assert inp.shape == spec.input_shape(), (inp.shape, spec.input_shape())
    assert filt.shape == spec.filter_shape(), (filt.shape, spec.filter_shape())
    cs = C.byref(spec.c_struct())
    if dst_type == DST_BITPACKED:
        thresholds = np.ascontiguousarray(thresholds, dtype=np.int32)
        out = np.empty(spec.output_shape(dst_type), np.int32)
        lib().lce_oracle_bconv2d_bitpacked(cs, _p(inp), _p(filt), _p(thresholds), _p(out), C.c_int(threads))
        return out
    mul, bias, cmin, cmax = fold_output_transform(spec, dst_type, post_mul, post_bias,
                                                  out_scale, out_zero_point)
    if dst_type == DST_F32:
        out = np.empty(spec.output_shape(dst_type), np.float32)
        cache = None
        if (spec.padding == PADDING_SAME and spec.pad_values == 0
                and spec.semantics == SEM_OPTIMIZED):
            cache = zero_pad_cache(spec, filt, post_mul)
        lib().lce_oracle_bconv2d_f32(cs, _p(inp), _p(filt), _p(mul), _p(bias), C.c_int32(cmin),
                                     C.c_int32(cmax), _p(cache) if cache is not None else None,
                                     _p(out), C.c_int(threads))
        return out
    out = np.empty(spec.output_shape(dst_type), np.int8)
    lib().lce_oracle_bconv2d_i8(cs, _p(inp), _p(filt), _p(mul), _p(bias), C.c_int32(cmin),
                                C.c_int32(cmax), _p(out), C.c_int(threads))
    return out


def bconv2d_indirect(spec: ConvSpec, dst_type: int, inp: np.ndarray, filt: np.ndarray, post_mul, post_bias,
                     out_scale: float = 1.0, out_zero_point: int = 0, threads: int = 1) -> np.ndarray:
    """The same op through the restatement of the reference's indirect BGEMM (packed 4-channel weight
    blocks, indirection table, 4x2 micro-kernel; core/indirect_bgemm/kernel.h, kernel_4x2_portable.h).
    Float / int8 output, VALID or one-padding."""
    inp = np.ascontiguousarray(inp, dtype=np.int32)
    filt = np.ascontiguousarray(filt, dtype=np.int32)
    assert inp.shape == spec.input_shape() and filt.shape == spec.filter_shape()
    mul, bias, cmin, cmax = fold_output_transform(spec, dst_type, post_mul, post_bias, out_scale, out_zero_point)
    out = np.empty(spec.output_shape(dst_type), np.float32 if dst_type == DST_F32 else np.int8)
    rc = lib().lce_oracle_bconv2d_indirect(C.byref(spec.c_struct()), _p(inp), _p(filt), C.c_int(dst_type), _p(mul),
                                           _p(bias), C.c_int32(cmin), C.c_int32(cmax), _p(out), C.c_int(threads))
    if rc != 0:
        raise ValueError(f"lce_oracle_bconv2d_indirect refused the case (rc {rc})")
    return out


def bmaxpool(x: np.ndarray, filter_h, filter_w, stride_h, stride_w, padding) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.int32)
    b, h, w, c = x.shape
    oh, ow = C.c_int32(), C.c_int32()
    args = (C.c_int32(b), C.c_int32(h), C.c_int32(w), C.c_int32(c), C.c_int32(filter_h),
            C.c_int32(filter_w), C.c_int32(stride_h), C.c_int32(stride_w), C.c_int32(padding))
    lib().lce_oracle_bmaxpool(*args, _p(x), C.byref(oh), C.byref(ow), None)
    out = np.empty((b, oh.value, ow.value, c), np.int32)
    lib().lce_oracle_bmaxpool(*args, _p(x), C.byref(oh), C.byref(ow), _p(out))
    return out
