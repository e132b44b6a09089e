"""Run the binary sections of a converted Larq model (.tflite) on the GPU, in true batches.

Host-side counterpart of the reference's Python ``Interpreter`` for the LCE custom ops
(larq_compute_engine/tflite/python/interpreter.py:58-98, interpreter_base.py:30-95): the same
property names and the same ``predict`` contract -- a NumPy array with an implicit leading batch
dimension (or a list of arrays, one per model input) in, concatenated predictions out -- but the
reference feeds samples one at a time through a batch-1 interpreter (the converter pins the batch
to 1, mlir/tf_tfl_passes.cc:141-144) while this runner re-plans every LceBconv2d for
``batch_size`` images and keeps all intermediate tensors in HBM.

Graphs made of LCE custom ops only (LceQuantize, LceBconv2d, LceBMaxPool2d, LceDequantize) run end to end
(``predict``).  A MIXED graph -- a real converted model: float stem, batch norms / adds between binary
convolutions, float head -- is cut into its binary SECTIONS (``Interpreter.sections``: the maximal groups of LCE ops
with no builtin operator between them, include/lce_tflite_model.h); ``run_section(k, inputs)`` runs one of them on
its boundary tensors, the float operators stay with TensorFlow Lite (``predict`` on such a graph raises
``NotImplementedError`` naming the first builtin operator).
The model file is read by the bounds-checked reader in csrc/tflite (include/lce_tflite_model.h).
"""
from __future__ import annotations

import ctypes as C
import importlib
import os
from typing import Dict, List, Optional, Sequence, Union

import numpy as np

_amd = importlib.import_module(__package__ or "compute-engine_amd")
_TFL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "tflite")
_tfl = None

FLOAT32, INT32, BOOL, INT8 = 0, 2, 6, 9
_NP = {FLOAT32: np.float32, INT32: np.int32, BOOL: np.bool_, INT8: np.int8}
LCE_OPS = ("LceQuantize", "LceDequantize", "LceBconv2d", "LceBMaxPool2d")


class _TensorInfo(C.Structure):
    _fields_ = [("type", C.c_int32), ("rank", C.c_int32), ("dims", C.c_int32 * 8), ("quantized", C.c_int32),
                ("scale", C.c_float), ("zero_point", C.c_int32), ("data", C.c_void_p), ("bytes", C.c_size_t),
                ("name", C.c_char_p)]


class _OperatorInfo(C.Structure):
    _fields_ = [("builtin_code", C.c_int32), ("custom_code", C.c_char_p), ("inputs", C.POINTER(C.c_int32)),
                ("num_inputs", C.c_int32), ("outputs", C.POINTER(C.c_int32)), ("num_outputs", C.c_int32),
                ("custom_options", C.POINTER(C.c_uint8)), ("custom_options_size", C.c_size_t)]


class _SectionInfo(C.Structure):
    _fields_ = [("ops", C.POINTER(C.c_int32)), ("num_ops", C.c_int32), ("inputs", C.POINTER(C.c_int32)), ("num_inputs", C.c_int32),
                ("outputs", C.POINTER(C.c_int32)), ("num_outputs", C.c_int32)]


class Section:
    """A maximal group of LCE ops with no builtin operator between them: operator indices in execution order, the
    non-constant tensors it reads from outside, the tensors it must deliver (read outside it, or graph outputs)."""

    def __init__(self, info: _SectionInfo):
        self.ops = [info.ops[i] for i in range(info.num_ops)]
        self.inputs = [info.inputs[i] for i in range(info.num_inputs)]
        self.outputs = [info.outputs[i] for i in range(info.num_outputs)]

    def __repr__(self):
        return f"Section(ops={self.ops}, inputs={self.inputs}, outputs={self.outputs})"


def tflite_lib() -> C.CDLL:
    global _tfl
    if _tfl is None:
        path = os.path.join(_TFL_DIR, "liblce_tflite_ops.so")
        if not os.path.exists(path):     # a build step belongs to the build, not to an import
            raise FileNotFoundError(f"{path} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    f"(or `make -C {_TFL_DIR}`)")
        l = C.CDLL(path)
        l.lce_tflite_model_open.restype = C.c_void_p
        l.lce_tflite_model_open.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        l.lce_tflite_model_close.argtypes = [C.c_void_p]
        for f in ("lce_tflite_model_num_tensors", "lce_tflite_model_num_operators"):
            getattr(l, f).argtypes = [C.c_void_p]
        for f in ("lce_tflite_model_inputs", "lce_tflite_model_outputs"):
            getattr(l, f).argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]
        l.lce_tflite_model_tensor.argtypes = [C.c_void_p, C.c_int32, C.POINTER(_TensorInfo)]
        l.lce_tflite_model_operator.argtypes = [C.c_void_p, C.c_int32, C.POINTER(_OperatorInfo)]
        l.lce_tflite_model_bconv2d_plan.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        l.lce_tflite_option_int.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_int32)]
        l.lce_tflite_model_last_error.restype = C.c_char_p
        l.lce_tflite_model_num_sections.argtypes = [C.c_void_p]
        l.lce_tflite_model_section.argtypes = [C.c_void_p, C.c_int32, C.POINTER(_SectionInfo)]
        l.lce_tflite_model_run_section.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p),
                                                   C.POINTER(C.c_void_p), C.c_void_p]
        l.lce_tflite_model_run_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]
        l.lce_tflite_model_run_stats.restype = None
        l.lce_tflite_model_use_hip_graphs.argtypes = [C.c_void_p, C.c_int32]
        l.lce_tflite_model_use_hip_graphs.restype = None
        l.lce_tflite_model_graph_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        l.lce_tflite_model_graph_stats.restype = None
        l.lce_tflite_model_section_tensor_shape.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                            C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]
        _tfl = l
    return _tfl


class Tensor:
    def __init__(self, info: _TensorInfo):
        self.type = info.type
        self.shape = tuple(info.dims[i] for i in range(info.rank))
        self.scale = float(info.scale) if info.quantized else None
        self.zero_point = int(info.zero_point) if info.quantized else None
        self.name = (info.name or b"").decode()
        self.constant = bool(info.data)


class Operator:
    def __init__(self, info: _OperatorInfo):
        self.builtin_code = info.builtin_code
        self.custom_code = (info.custom_code or b"").decode()
        self.inputs = [info.inputs[i] for i in range(info.num_inputs)]
        self.outputs = [info.outputs[i] for i in range(info.num_outputs)]
        self._opts = (info.custom_options, info.custom_options_size)

    def option(self, key: str) -> Optional[int]:
        v = C.c_int32()
        rc = tflite_lib().lce_tflite_option_int(C.cast(self._opts[0], C.c_void_p), self._opts[1], key.encode(), C.byref(v))
        return None if rc else v.value


class LceModel:
    """A parsed .tflite flatbuffer (first subgraph)."""

    def __init__(self, flatbuffer: Union[bytes, str, os.PathLike]):
        if not isinstance(flatbuffer, (bytes, bytearray)):
            with open(flatbuffer, "rb") as f:
                flatbuffer = f.read()
        self._data = bytes(flatbuffer)            # must outlive the handle (zero-copy reader)
        err = C.create_string_buffer(256)
        self._h = tflite_lib().lce_tflite_model_open(self._data, len(self._data), err, 256)
        if not self._h:
            raise ValueError("not a readable TFLite model: " + err.value.decode(errors="replace"))
        l = tflite_lib()
        self.tensors: List[Tensor] = []
        for i in range(l.lce_tflite_model_num_tensors(self._h)):
            info = _TensorInfo()
            _amd.check(l.lce_tflite_model_tensor(self._h, i, C.byref(info)))
            self.tensors.append(Tensor(info))
        self.operators: List[Operator] = []
        for i in range(l.lce_tflite_model_num_operators(self._h)):
            info = _OperatorInfo()
            _amd.check(l.lce_tflite_model_operator(self._h, i, C.byref(info)))
            self.operators.append(Operator(info))
        buf = (C.c_int32 * 64)()
        self.inputs = [buf[i] for i in range(l.lce_tflite_model_inputs(self._h, buf, 64))]
        self.outputs = [buf[i] for i in range(l.lce_tflite_model_outputs(self._h, buf, 64))]
        self.sections: List[Section] = []
        for i in range(l.lce_tflite_model_num_sections(self._h)):
            info = _SectionInfo()
            _amd.check(l.lce_tflite_model_section(self._h, i, C.byref(info)))
            self.sections.append(Section(info))

    def close(self):
        if getattr(self, "_h", None):
            tflite_lib().lce_tflite_model_close(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc: int):
        if rc:
            msg = tflite_lib().lce_tflite_model_last_error().decode() or _amd.lib().lce_hip_last_error().decode()
            raise _amd.LceHipError(rc, msg)

    def section_tensor_shape(self, section: int, tensor: int, batch: int, semantics: int = _amd.SEM_OPTIMIZED):
        """([N, H, W, C], bytes) of a tensor the section reads or produces at ``batch`` images (C in words when bitpacked)."""
        dims, nbytes = (C.c_int32 * 4)(), C.c_size_t()
        self._check(tflite_lib().lce_tflite_model_section_tensor_shape(self._h, section, tensor, batch, semantics, dims, C.byref(nbytes)))
        return tuple(dims), int(nbytes.value)

    def run_section(self, section: int, batch: int, inputs_dev: Sequence[int], outputs_dev: Sequence[int], stream: int = 0,
                    semantics: int = _amd.SEM_OPTIMIZED):
        """``lce_tflite_model_run_section`` (include/lce_tflite_model.h) on raw device pointers."""
        i = (C.c_void_p * max(1, len(inputs_dev)))(*inputs_dev)
        o = (C.c_void_p * max(1, len(outputs_dev)))(*outputs_dev)
        self._check(tflite_lib().lce_tflite_model_run_section(self._h, section, batch, semantics, i, o, C.c_void_p(stream)))

    def run_stats(self):
        """(plans cached in the model, LceQuantize ops the last run folded into a convolution, bytes of intermediates)."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_size_t()
        tflite_lib().lce_tflite_model_run_stats(self._h, C.byref(a), C.byref(b), C.byref(c))
        return int(a.value), int(b.value), int(c.value)

    def use_hip_graphs(self, on: bool = True):
        """``lce_tflite_model_use_hip_graphs``: run_section records a section's launches once per (batch, stream, tensor
        pointers) and replays them as one launch; needs a stream of its own (not the null stream)."""
        tflite_lib().lce_tflite_model_use_hip_graphs(self._h, 1 if on else 0)

    def graph_stats(self):
        """(recordings made, run_section calls served by one)."""
        a, b = C.c_int32(), C.c_int32()
        tflite_lib().lce_tflite_model_graph_stats(self._h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def bconv2d_plan(self, op_index: int, batch: int, semantics: int = _amd.SEM_OPTIMIZED) -> "_amd.Bconv2dPlan":
        """A ready plan (weights set) for LceBconv2d operator ``op_index`` at the given batch size."""
        h = C.c_void_p()
        rc = tflite_lib().lce_tflite_model_bconv2d_plan(self._h, op_index, batch, semantics, C.byref(h))
        if rc:
            msg = tflite_lib().lce_tflite_model_last_error().decode() or _amd.lib().lce_hip_last_error().decode()
            raise _amd.LceHipError(rc, msg)
        out_type = self.tensors[self.operators[op_index].outputs[0]].type
        dst = {FLOAT32: _amd.F32, INT8: _amd.I8, INT32: _amd.BITPACKED}[out_type]
        cout = self.tensors[self.operators[op_index].inputs[1]].shape[0]
        return _amd.Bconv2dPlan.from_handle(h.value, dst, cout)


class Interpreter:
    """``Interpreter(flatbuffer_model, batch_size=...)`` -- see the module docstring."""

    def __init__(self, flatbuffer_model, batch_size: int = 256, device: str = "cuda:0",
                 use_reference_bconv: bool = False):
        self.model = flatbuffer_model if isinstance(flatbuffer_model, LceModel) else LceModel(flatbuffer_model)
        self.batch_size = int(batch_size)
        self.device = device
        self._sem = _amd.SEM_REFERENCE if use_reference_bconv else _amd.SEM_OPTIMIZED
        self._foreign = [(i, op) for i, op in enumerate(self.model.operators)
                         if op.builtin_code != 32 or op.custom_code not in LCE_OPS]

    @property
    def sections(self) -> List[Section]:
        """The binary sections of the graph (one covering everything for an LCE-only graph)."""
        return self.model.sections

    @property
    def lce_only(self) -> bool:
        return not self._foreign

    def _require_lce_only(self):
        if self._foreign:
            i, op = self._foreign[0]
            raise NotImplementedError(
                "only LCE custom ops run here; the model contains builtin operator %d %r (operator %d): run its %d binary "
                "section(s) with run_section() and keep the float operators in TensorFlow Lite"
                % (op.builtin_code, op.custom_code, i, len(self.model.sections)))

    # ---- the reference Interpreter's properties (interpreter_base.py:34-72) -------------------
    def _props(self, ids):
        t = [self.model.tensors[i] for i in ids]
        return t

    @property
    def input_types(self):
        return [_NP[t.type] for t in self._props(self.model.inputs)]

    @property
    def input_shapes(self):
        return [t.shape for t in self._props(self.model.inputs)]

    @property
    def input_scales(self):
        return [t.scale for t in self._props(self.model.inputs)]

    @property
    def input_zero_points(self):
        return [t.zero_point for t in self._props(self.model.inputs)]

    @property
    def output_types(self):
        return [_NP[t.type] for t in self._props(self.model.outputs)]

    @property
    def output_shapes(self):
        return [t.shape for t in self._props(self.model.outputs)]

    @property
    def output_scales(self):
        return [t.scale for t in self._props(self.model.outputs)]

    @property
    def output_zero_points(self):
        return [t.zero_point for t in self._props(self.model.outputs)]

    # ---- execution: the C entry lce_tflite_model_run_section does the work (plans cached per batch size in the model,
    # LceBconv2d + LceQuantize fused through run_dual, intermediates in buffers the model owns) -------------------------
    def _run_section_device(self, index: int, live, batch: int, wanted=None):
        """Section `index` on CUDA tensors: `live` maps tensor index -> tensor for every entry of ``sections[index].inputs``;
        returns the tensors `wanted` (default: the section's outputs) as new CUDA tensors.  Asynchronous on the current
        torch stream."""
        import torch
        sec = self.model.sections[index]
        dt = {FLOAT32: torch.float32, INT32: torch.int32, BOOL: torch.bool, INT8: torch.int8}
        outs = []
        for t in sec.outputs:
            dims, _ = self.model.section_tensor_shape(index, t, batch, self._sem)
            outs.append(torch.empty(dims, dtype=dt[self.model.tensors[t].type], device=self.device))
        ins = [live[t].contiguous() for t in sec.inputs]
        with torch.cuda.device(torch.device(self.device)):
            stream = torch.cuda.current_stream().cuda_stream
            self.model.run_section(index, batch, [x.data_ptr() for x in ins], [y.data_ptr() for y in outs], stream, self._sem)
            for x in ins:                                    # (the launches read them after this call returns)
                x.record_stream(torch.cuda.current_stream())
        by_index = dict(zip(sec.outputs, outs))
        return [by_index[t] for t in (sec.outputs if wanted is None else wanted)]

    def _run_ops(self, live, batch):
        """The whole graph (LCE ops only: ONE section) on device tensors; returns the graph outputs."""
        return self._run_section_device(0, live, batch, self.model.outputs)

    def run_section(self, index: int, inputs: Union[Sequence[np.ndarray], Dict[int, np.ndarray]]):
        """Runs binary section `index` of a (mixed) graph on its boundary tensors: `inputs` = one array per entry of
        ``sections[index].inputs`` in that order (or a dict tensor index -> array), each with a leading batch axis of any
        size (the plans are re-made for it); returns one NumPy array per entry of ``sections[index].outputs``.  Everything in
        between stays in HBM.  The float operators around the section are TensorFlow Lite's job."""
        import torch
        sec = self.model.sections[index]
        if isinstance(inputs, dict):
            arrs = [inputs[t] for t in sec.inputs]
        else:
            arrs = list(inputs)
        if len(arrs) != len(sec.inputs):
            raise ValueError("section %d reads %d tensor(s) %r, got %d array(s)" % (index, len(sec.inputs), sec.inputs, len(arrs)))
        live = {}
        for t, a in zip(sec.inputs, arrs):
            info = self.model.tensors[t]
            a = np.ascontiguousarray(a, dtype=_NP[info.type])
            if tuple(a.shape[1:]) != tuple(info.shape[1:]):
                raise ValueError("tensor %d (%s) has shape [batch, %s], got %r" % (t, info.name, ", ".join(map(str, info.shape[1:])), a.shape))
            live[t] = torch.from_numpy(a).to(self.device)
        batch = arrs[0].shape[0]
        if any(a.shape[0] != batch for a in arrs):
            raise ValueError("all inputs of a section share the batch dimension")
        return [y.cpu().numpy() for y in self._run_section_device(index, live, batch)]

    def _run_batch(self, inputs):
        """One batch, synchronously (kept for callers that drive batches themselves)."""
        import torch
        self._require_lce_only()
        live = {idx: torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
                for idx, arr in zip(self.model.inputs, inputs)}
        return [y.cpu().numpy() for y in self._run_ops(live, inputs[0].shape[0])]

    def _batches(self, x):
        """The reference's input forms (interpreter_base.py:10-27) -- an array with a leading sample axis, a
        list of such arrays (one per model input), or an iterator yielding ONE sample at a time (an array, or
        a list of arrays for several inputs) -- regrouped into batches of ``batch_size`` samples."""
        n_in = len(self.model.inputs)
        if isinstance(x, np.ndarray):
            x = [x]
        if isinstance(x, (list, tuple)):
            xs = list(x)
            if not xs or len(xs) != n_in:
                raise ValueError("expected one array per model input (%d)" % n_in)
            for b0 in range(0, xs[0].shape[0], self.batch_size):
                yield [a[b0:b0 + self.batch_size] for a in xs]
        elif hasattr(x, "__next__") and hasattr(x, "__iter__"):
            pending = [[] for _ in range(n_in)]
            for sample in x:
                parts = [sample] if isinstance(sample, np.ndarray) else list(sample)
                if len(parts) != n_in:
                    raise ValueError("expected one array per model input (%d)" % n_in)
                for p, a in zip(pending, parts):
                    p.append(a)
                if len(pending[0]) == self.batch_size:
                    yield [np.stack(p) for p in pending]
                    pending = [[] for _ in range(n_in)]
            if pending[0]:
                yield [np.stack(p) for p in pending]
        else:
            raise ValueError("Expected either a list of inputs or a Numpy array with implicit initial batch dimension "
                             "or an iterator yielding one of the above. Received: %r" % (x,))

    def predict(self, x, verbose: int = 0):
        """Input samples -> concatenated predictions (interpreter_base.py:74-95), ``batch_size`` samples per
        device pass instead of the reference's one.  Batches flow through a three-stage pipeline on three
        streams with page-locked staging buffers -- H2D of batch k+1 | the LCE ops of batch k | D2H of
        batch k-1 -- so the PCIe copies (40x the kernel time for a float 56x56x256 map) overlap the compute
        and each other's direction."""
        import torch
        self._require_lce_only()
        dev = torch.device(self.device)
        n_in = len(self.model.inputs)
        in_dt = [self.input_types[k] for k in range(n_in)]
        with torch.cuda.device(dev):
            s_in, s_out, s_run = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.current_stream(dev)
            pin_in = [[None] * n_in for _ in range(2)]       # two sets of page-locked input staging buffers
            dev_in = [[None] * n_in for _ in range(2)]
            in_free = [None, None]                           # event: the ops that read dev_in[slot] have finished
            flights = []                                     # (pinned outputs, event) of the batches not yet collected
            results = None

            def collect(flight):
                nonlocal results
                outs, ev = flight
                ev.synchronize()
                arrs = [o.numpy().copy() for o in outs]
                results = [[a] for a in arrs] if results is None else [r + [a] for r, a in zip(results, arrs)]

            for k, batch in enumerate(self._batches(x)):
                slot, b = k & 1, batch[0].shape[0]
                if in_free[slot] is not None:
                    s_in.wait_event(in_free[slot])           # batch k-2's ops no longer read this slot's tensors
                    in_free[slot].synchronize()              # ... and its staging buffer has been copied out of
                live = {}
                for j, (idx, arr) in enumerate(zip(self.model.inputs, batch)):
                    arr = np.ascontiguousarray(arr, dtype=in_dt[j])
                    if pin_in[slot][j] is None or pin_in[slot][j].shape[0] < b or pin_in[slot][j].shape[1:] != arr.shape[1:]:
                        cap = (max(b, self.batch_size),) + arr.shape[1:]
                        pin_in[slot][j] = torch.empty(cap, dtype=torch.from_numpy(arr[:0]).dtype).pin_memory()
                        dev_in[slot][j] = torch.empty(cap, dtype=pin_in[slot][j].dtype, device=dev)
                    pin_in[slot][j][:b].copy_(torch.from_numpy(arr))
                    with torch.cuda.stream(s_in):
                        dev_in[slot][j][:b].copy_(pin_in[slot][j][:b], non_blocking=True)
                    live[idx] = dev_in[slot][j][:b]
                s_run.wait_stream(s_in)
                outs_dev = self._run_ops(live, b)
                done = torch.cuda.Event()
                done.record(s_run)
                in_free[slot] = done
                s_out.wait_event(done)
                outs_pin = []
                with torch.cuda.stream(s_out):
                    for y in outs_dev:
                        y.record_stream(s_out)
                        hp = torch.empty(y.shape, dtype=y.dtype).pin_memory()
                        hp.copy_(y, non_blocking=True)
                        outs_pin.append(hp)
                    ev = torch.cuda.Event()
                    ev.record(s_out)
                flights.append((outs_pin, ev))
                if len(flights) > 1:
                    collect(flights.pop(0))                  # batch k-1: its D2H overlapped this batch's H2D + ops
            for f in flights:
                collect(f)
        if results is None:
            raise ValueError("predict() needs at least one sample")
        outputs = [np.concatenate(o) for o in results]
        return outputs[0] if len(self.model.outputs) == 1 else outputs
