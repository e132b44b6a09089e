#!/usr/bin/env python
"""lce_tflite_model_run_section, eager vs recorded HIP graph (lce_tflite_model_use_hip_graphs): wall time per call of a binary
section made of QuickNet's short layers -- float in -> LceQuantize -> N x (LceBconv2d 3x3 float -> LceQuantize) -- at batch 256.
usage: section_graph_ab.py [calls]      (C ABI only: lce_hip_malloc / streams; the model file is written by tests/tflite_writer.py)"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
amd = importlib.import_module("compute-engine_amd")
mr = importlib.import_module("compute-engine_amd.model_runner")
import oracle_lib as O           # noqa: E402  (shapes / option maps only)
import synth                     # noqa: E402
from tflite_writer import ModelBuilder            # noqa: E402
from test_model_reader_host import bconv_options  # noqa: E402


def chain_model(hw, c, layers, seed):
    b = ModelBuilder()
    t_in = b.tensor([1, hw, hw, c], np.float32, "input")
    t = b.tensor([1, hw, hw, c // 32], np.int32, "q0")
    b.custom_op("LceQuantize", [t_in], [t], b"")
    outs = []
    for k in range(layers):
        s = O.ConvSpec(1, hw, hw, c, 3, 3, c, padding=O.PADDING_SAME, pad_values=1)
        _, w, m, bias = synth.conv_inputs(s, seed + k)
        tw, tm, tb = b.tensor(w.shape, np.int32, f"w{k}", w), b.tensor([c], np.float32, f"m{k}", m), b.tensor([c], np.float32, f"b{k}", bias)
        ty = b.tensor([1, hw, hw, c], np.float32, f"y{k}")
        b.custom_op("LceBconv2d", [t, tw, tm, tb, -1], [ty], bconv_options(s))
        outs.append(ty)
        if k + 1 < layers:
            t = b.tensor([1, hw, hw, c // 32], np.int32, f"q{k + 1}")
            b.custom_op("LceQuantize", [ty], [t], b"")
    b.inputs, b.outputs = [t_in], [outs[-1]]
    return b.finish()


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    lib = amd.lib()
    stream = C.c_void_p()
    amd.check(lib.lce_hip_stream_create(C.byref(stream)))
    for hw, c, layers in ((7, 512, 4), (14, 256, 4), (7, 512, 8)):
        model = mr.LceModel(chain_model(hw, c, layers, 100 + hw))
        sec = model.sections[0]
        n = 256
        ins, outs = [], []
        for t in sec.inputs:
            _, nbytes = model.section_tensor_shape(0, t, n)
            d = C.c_void_p()
            amd.check(lib.lce_hip_malloc(C.byref(d), C.c_size_t(nbytes)))
            amd.check(lib.lce_hip_memset(d, 0x3C, C.c_size_t(nbytes), stream))
            ins.append(d.value)
        for t in sec.outputs:
            _, nbytes = model.section_tensor_shape(0, t, n)
            d = C.c_void_p()
            amd.check(lib.lce_hip_malloc(C.byref(d), C.c_size_t(nbytes)))
            outs.append(d.value)
        res = {}
        for mode in ("eager", "graph", "eager", "graph"):
            model.use_hip_graphs(mode == "graph")
            for _ in range(20):
                model.run_section(0, n, ins, outs, stream=stream.value)
            amd.check(lib.lce_hip_stream_synchronize(stream))
            t0 = time.perf_counter()
            for _ in range(calls):
                model.run_section(0, n, ins, outs, stream=stream.value)
            amd.check(lib.lce_hip_stream_synchronize(stream))
            res.setdefault(mode, []).append((time.perf_counter() - t0) / calls * 1e3)
        rec, rep = model.graph_stats()
        print(f"{layers} x LceBconv2d 3x3 {hw}x{hw}x{c} (+ LceQuantize between), batch {n}: eager {min(res['eager']):.4f} ms per call, "
              f"recorded graph {min(res['graph']):.4f} ms  ({len(sec.ops)} operators, {model.run_stats()[1]} LceQuantize fused; {rec} recordings, {rep} replays)")
        for d in ins + outs:
            amd.check(lib.lce_hip_free(C.c_void_p(d)))
        model.close()
    amd.check(lib.lce_hip_stream_destroy(stream))


if __name__ == "__main__":
    main()
