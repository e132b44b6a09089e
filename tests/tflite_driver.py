"""ctypes front-end of the single-op driver in compute-engine_amd/csrc/tflite/ -- the
stand-in for TFLite's SingleOpModel that the reference's op tests use
(tflite/tests/bconv2d_op_model.h:24-59)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "compute-engine_amd", "csrc", "tflite")
_lib = None

# TfLiteType / TfLiteAllocationType values (tensorflow/lite/core/c/c_api_types.h, common.h)
FLOAT32, INT32, BOOL, INT8 = 1, 2, 6, 9
MMAP_RO, ARENA_RW, DYNAMIC = 1, 2, 4
_NP = {FLOAT32: np.float32, INT32: np.int32, BOOL: np.bool_, INT8: np.int8}
BCONV_DEFAULT, BCONV_REF, BCONV_OPT_BGEMM, BCONV_OPT_INDIRECT = 0, 1, 2, 3


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_DIR, "liblce_tflite_ops.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", _DIR], check=True, capture_output=True)
        l = C.CDLL(path)
        l.lce_driver_create.restype = C.c_void_p
        l.lce_driver_create.argtypes = [C.c_char_p, C.c_int, C.c_int]
        l.lce_driver_forget_graph.argtypes = [C.c_void_p]
        l.lce_driver_forget_graph.restype = None
        l.lce_driver_destroy.argtypes = [C.c_void_p]
        l.lce_driver_add_tensor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int,
                                            C.c_float, C.c_int, C.c_int]
        l.lce_driver_set_data.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        l.lce_driver_set_node.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int,
                                          C.c_char_p, C.c_size_t]
        l.lce_driver_prepare.argtypes = [C.c_void_p]
        l.lce_driver_invoke.argtypes = [C.c_void_p]
        for f in ("lce_driver_tensor_rank", "lce_driver_num_temporaries"):
            getattr(l, f).argtypes = [C.c_void_p] + ([C.c_int] if "rank" in f else [])
        l.lce_driver_tensor_dim.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.lce_driver_tensor_bytes.argtypes = [C.c_void_p, C.c_int]
        l.lce_driver_tensor_bytes.restype = C.c_size_t
        l.lce_driver_get_data.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        l.lce_driver_log.argtypes = [C.c_void_p]
        l.lce_driver_log.restype = C.c_char_p
        l.lce_driver_flex_lookup.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.lce_driver_create_chain.restype = C.c_void_p
        l.lce_driver_add_node.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int,
                                          C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_size_t]
        l.lce_driver_prepare_all.argtypes = [C.c_void_p]
        l.lce_driver_invoke_all.argtypes = [C.c_void_p]
        l.lce_tflite_ops_set_residency.argtypes = [C.c_int]
        l.lce_driver_declare_graph.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        l.lce_driver_declare_graph.restype = None
        l.lce_driver_forget_graph.argtypes = [C.c_void_p]
        l.lce_driver_forget_graph.restype = None
        l.lce_tflite_ops_device_buffers.argtypes = [C.POINTER(C.c_uint64)] * 2
        l.lce_tflite_ops_device_buffers.restype = None
        l.lce_tflite_ops_transfer_counts.argtypes = [C.POINTER(C.c_uint64)] * 4 + [C.c_int]
        _lib = l
    return _lib


def flex_lookup(buf: bytes, key: str):
    null, val = C.c_int(), C.c_int()
    rc = lib().lce_driver_flex_lookup(buf, len(buf), key.encode(), C.byref(null), C.byref(val))
    if rc:
        raise ValueError("not a flexbuffer map")
    return bool(null.value), val.value


class SingleOpModel:
    def __init__(self, op_name: str, variant: int = 0, use_resolver: bool = False):
        self._keep = []
        self._h = lib().lce_driver_create(op_name.encode(), variant, int(use_resolver))
        if not self._h:
            raise ValueError(f"unknown op {op_name}")
        self._dtype = {}

    def close(self):
        if self._h:
            lib().lce_driver_destroy(self._h)
            self._h = None

    __del__ = close

    def add_tensor(self, ttype, shape, data=None, allocation=ARENA_RW, scale=0.0, zero_point=0, affine=False):
        dims = (C.c_int * len(shape))(*shape)
        idx = lib().lce_driver_add_tensor(self._h, ttype, len(shape), dims, allocation, float(scale),
                                          int(zero_point), int(affine))
        self._dtype[idx] = _NP[ttype]
        if data is not None:
            self.set_data(idx, data)
        return idx

    def set_data(self, idx, data):
        a = np.ascontiguousarray(data, self._dtype[idx])
        assert lib().lce_driver_set_data(self._h, idx, a.ctypes.data_as(C.c_void_p), a.nbytes) == 0

    def set_node(self, inputs, outputs, options: bytes = b""):
        i = (C.c_int * len(inputs))(*inputs)
        o = (C.c_int * len(outputs))(*outputs)
        self._keep.append(options)
        lib().lce_driver_set_node(self._h, i, len(inputs), o, len(outputs), options, len(options))

    def prepare(self) -> int:
        return lib().lce_driver_prepare(self._h)

    def invoke(self) -> int:
        return lib().lce_driver_invoke(self._h)

    def shape(self, idx):
        return tuple(lib().lce_driver_tensor_dim(self._h, idx, d) for d in range(lib().lce_driver_tensor_rank(self._h, idx)))

    def get(self, idx):
        out = np.empty(self.shape(idx), self._dtype[idx])
        assert lib().lce_driver_get_data(self._h, idx, out.ctypes.data_as(C.c_void_p), out.nbytes) == 0
        return out

    @property
    def log(self) -> str:
        return lib().lce_driver_log(self._h).decode(errors="replace")

    @property
    def num_temporaries(self) -> int:
        return lib().lce_driver_num_temporaries(self._h)


class ChainModel(SingleOpModel):
    """Several nodes in execution order behind ONE context.  As in TensorFlow Lite, the context's GetExecutionPlan /
    GetNodeAndRegistration are forbidden to kernels (they log an error and fail); `declare_graph` is what an application
    does once with its interpreter so that tensors only LCE ops read stay on the device.  "HostCopy" is a stand-in for a
    builtin CPU kernel (copies its input to its output in the arena)."""

    def __init__(self):
        self._keep = []
        self._h = lib().lce_driver_create_chain()
        self._dtype = {}

    def add_node(self, op_name, inputs, outputs, options: bytes = b"", variant: int = 0, use_resolver: bool = True) -> int:
        i = (C.c_int * len(inputs))(*inputs)
        o = (C.c_int * len(outputs))(*outputs)
        idx = lib().lce_driver_add_node(self._h, op_name.encode(), variant, int(use_resolver), i, len(inputs), o, len(outputs),
                                        options, len(options))
        if idx < 0:
            raise ValueError(f"unknown op {op_name}")
        return idx

    def declare_graph(self, graph_outputs):
        """lce_tflite_ops_declare_graph_* as lce_ops_register.h's DeclareGraphForDeviceResidency makes them."""
        o = (C.c_int * len(graph_outputs))(*graph_outputs)
        lib().lce_driver_declare_graph(self._h, o, len(graph_outputs))

    def forget_graph(self):
        lib().lce_driver_forget_graph(self._h)

    def prepare(self) -> int:
        return lib().lce_driver_prepare_all(self._h)

    def invoke(self) -> int:
        return lib().lce_driver_invoke_all(self._h)


def transfer_counts(reset: bool = False):
    """(host->device copies, device->host copies, bytes up, bytes down) made by the ops' invokes since the last reset."""
    v = [C.c_uint64() for _ in range(4)]
    lib().lce_tflite_ops_transfer_counts(*[C.byref(x) for x in v], int(reset))
    return tuple(int(x.value) for x in v)


def device_buffers():
    """(count, bytes) of the device buffers the ops' residency layer holds right now, over all contexts."""
    n, b = C.c_uint64(), C.c_uint64()
    lib().lce_tflite_ops_device_buffers(C.byref(n), C.byref(b))
    return int(n.value), int(b.value)


def set_residency(on: bool):
    lib().lce_tflite_ops_set_residency(int(on))


def build_bconv2d(spec, dst, filt, post_mul, post_bias, thresholds, variant=BCONV_DEFAULT,
                  out_scale=1.0, out_zero_point=0, use_resolver=False, options=None, in_alloc=ARENA_RW):
    """Mirror of BConv2DOpModel (tflite/tests/bconv2d_op_model.h): 5 inputs, 1 output."""
    import flexbuf
    import oracle_lib as O
    m = SingleOpModel("LceBconv2d", variant, use_resolver)
    x = m.add_tensor(INT32, spec.input_shape(), allocation=in_alloc)
    f = m.add_tensor(INT32, spec.filter_shape(), filt, allocation=MMAP_RO)
    if dst == O.DST_BITPACKED:
        pm, pb = -1, -1
        th = m.add_tensor(INT32, (spec.channels_out,), thresholds, allocation=MMAP_RO)
    else:
        pm = m.add_tensor(FLOAT32, (spec.channels_out,), post_mul, allocation=MMAP_RO)
        pb = m.add_tensor(FLOAT32, (spec.channels_out,), post_bias, allocation=MMAP_RO)
        th = -1
    ttype = {O.DST_F32: FLOAT32, O.DST_I8: INT8, O.DST_BITPACKED: INT32}[dst]
    out = m.add_tensor(ttype, (), scale=out_scale, zero_point=out_zero_point, affine=(dst == O.DST_I8))
    if options is None:
        options = flexbuf.bconv2d_options(spec.channels_in, spec.stride_h, spec.stride_w, spec.dilation_h,
                                          spec.dilation_w, spec.padding, spec.pad_values, spec.activation)
    m.set_node([x, f, pm, pb, th], [out], options)
    return m, x, out
