"""The batched model runner (compute-engine_amd/model_runner.py) on the
GPU: a .tflite made of LCE ops is read, every LceBconv2d is planned from the file, and
``Interpreter.predict`` returns what the CPU oracle computes op by op -- in true batches, with a
ragged last batch, and identically to the reference's one-sample-at-a-time contract."""
import importlib

import numpy as np
import pytest

import synth
import oracle_lib as O
from test_model_reader_host import mixed_model, oracle_forward, small_model

amd = importlib.import_module("compute-engine_amd")
mr = importlib.import_module("compute-engine_amd.model_runner")

pytestmark = pytest.mark.gpu


def test_interpreter_predict_matches_oracle_in_true_batches():
    data, p = small_model(21)
    x = synth.rng(5).uniform(-1.5, 1.5, (37, 12, 12, 64)).astype(np.float32)
    want_i8, want_deq = oracle_forward(x, p)
    it = mr.Interpreter(data, batch_size=16)
    assert it.input_shapes == [(1, 12, 12, 64)] and it.input_types == [np.float32]
    assert it.output_types == [np.int8, np.float32]
    assert it.output_scales[0] == pytest.approx(float(p["q3"][0])) and it.output_zero_points[0] == p["q3"][1]
    got_i8, got_deq = it.predict(x)
    assert got_i8.shape == want_i8.shape and np.array_equal(got_i8, want_i8)
    assert np.array_equal(got_deq, want_deq)
    # the float LceBconv2d and the LceQuantize of its output ran as ONE pass (second output of the epilogue); plans were
    # made once per convolution and distinct batch size (16 and the ragged 5) and live in the model
    n_convs = sum(op.custom_code == "LceBconv2d" for op in it.model.operators)
    plans, fused, scratch = it.model.run_stats()
    assert fused == 1 and plans == 2 * n_convs and scratch > 0
    it.predict(x)
    assert it.model.run_stats()[0] == plans
    # the reference's contract (interpreter_base.py:74-95): sample by sample gives the same answer
    one = mr.Interpreter(data, batch_size=1)
    a, b = one.predict(x[:3])
    assert np.array_equal(a, want_i8[:3]) and np.array_equal(b, want_deq[:3])
    # a list with one array per model input is accepted too
    c, _ = it.predict([x[:4]])
    assert np.array_equal(c, want_i8[:4])


def test_reference_semantics_flag():
    """use_reference_bconv follows Register_BCONV_2D_REF (exact SAME-zero padding)."""
    data, _ = small_model(22)
    mr.Interpreter(data, batch_size=4, use_reference_bconv=True).predict(
        synth.rng(6).uniform(-1, 1, (4, 12, 12, 64)).astype(np.float32))


def test_predict_accepts_the_reference_iterator_forms_and_pipelines_many_batches():
    """interpreter_base.py:10-27: predict takes an array, a list of arrays, or an ITERATOR that yields one
    sample at a time (an array without the sample axis, or a list of such for several inputs).  Seven batches
    flow through the H2D | ops | D2H pipeline (two staging slots reused three times each) and come back in
    order."""
    data, p = small_model(23)
    x = synth.rng(8).uniform(-1.5, 1.5, (100, 12, 12, 64)).astype(np.float32)
    want_i8, want_deq = oracle_forward(x, p)
    it = mr.Interpreter(data, batch_size=16)
    for form in (iter(x), iter([s] for s in x), (s for s in x)):
        got_i8, got_deq = it.predict(form)
        assert np.array_equal(got_i8, want_i8) and np.array_equal(got_deq, want_deq)
    with pytest.raises(ValueError):
        it.predict("not samples")
    with pytest.raises(ValueError):
        it.predict(iter(()))


def test_binary_sections_of_a_mixed_graph_run_on_their_boundary_tensors():
    """A QuickNet-shaped mixed graph (builtin CONV_2D stem, ADDs between LceBconv2ds, builtin pooling head): every binary
    section runs on the GPU from its boundary tensors, in true batches, and equals the oracle run op by op; the float
    operators in between are done here in NumPy (TensorFlow Lite's job in a deployment)."""
    data, t, p = mixed_model(31)
    it = mr.Interpreter(data, batch_size=8)
    n = 5
    g = synth.rng(9)
    stem = g.standard_normal((n, 10, 10, 64)).astype(np.float32)          # what the float stem would hand over
    s_a, _, s_c, s_d = (s.with_batch(n) for s in p["specs"])
    # section 0: LceQuantize + LceBconv2d (one pass through run_dual would need a reader of the bits; here: two ops)
    (y0,) = it.run_section(0, [stem])
    want_y0 = O.bconv2d(s_a, O.DST_F32, O.bitpack(stem), p["w"][0], p["m"][0], p["b"][0])
    assert np.array_equal(y0.view(np.int32), want_y0.view(np.int32))
    r0 = y0 + stem                                                           # builtin ADD
    (y1,) = it.run_section(1, {t["r0"]: r0})
    want_y1 = O.bconv2d(s_a, O.DST_F32, O.bitpack(r0), p["w"][1], p["m"][1], p["b"][1])
    assert np.array_equal(y1.view(np.int32), want_y1.view(np.int32))
    r1 = y1 + r0
    outs = dict(zip(it.sections[2].outputs, it.run_section(2, [r1])))
    b2 = O.bconv2d(s_c, O.DST_BITPACKED, O.bitpack(r1), p["w"][2], thresholds=p["thr2"])
    want_y3 = O.bconv2d(s_d, O.DST_F32, O.bmaxpool(b2, 2, 2, 2, 2, O.PADDING_VALID), p["w"][3], p["m"][3], p["b"][3])
    assert np.array_equal(outs[t["y3"]].view(np.int32), want_y3.view(np.int32))
    assert np.array_equal(outs[t["d2"]], O.unpack(b2, 96, np.float32))
    with pytest.raises(ValueError, match="reads 1 tensor"):
        it.run_section(2, [r1, r1])
    with pytest.raises(ValueError, match="has shape"):
        it.run_section(2, [r1[:, :5]])


def test_run_section_through_the_c_abi_alone():
    """lce_tflite_model_run_section as a C host calls it: device memory from lce_hip_malloc, copies through lce_hip_memcpy_*,
    no torch anywhere -- the route of examples/lce_minimal.cc:28-62 for the binary part of a graph -- against the oracle
    run op by op, at two batch sizes on one model (plans cached per batch)."""
    import ctypes as C
    data, p = small_model(24)
    model = mr.LceModel(data)
    assert len(model.sections) == 1
    sec = model.sections[0]
    lib = amd.lib()
    for n in (6, 3):
        x = synth.rng(30 + n).uniform(-1.5, 1.5, (n, 12, 12, 64)).astype(np.float32)
        want = dict(zip(model.outputs, oracle_forward(x, p)))
        ins, outs, host = [], [], []
        for t in sec.inputs:
            dims, nbytes = model.section_tensor_shape(0, t, n)
            assert dims == x.shape and nbytes == x.nbytes
            d = C.c_void_p()
            amd.check(lib.lce_hip_malloc(C.byref(d), C.c_size_t(nbytes)))
            amd.check(lib.lce_hip_memcpy_h2d(d, x.ctypes.data_as(C.c_void_p), C.c_size_t(nbytes), None))
            ins.append(d.value)
        for t in sec.outputs:
            dims, nbytes = model.section_tensor_shape(0, t, n)
            d = C.c_void_p()
            amd.check(lib.lce_hip_malloc(C.byref(d), C.c_size_t(nbytes)))
            outs.append(d.value)
            host.append(np.empty(dims, mr._NP[model.tensors[t].type]))
            assert host[-1].nbytes == nbytes
        model.run_section(0, n, ins, outs)
        for t, d, h in zip(sec.outputs, outs, host):
            amd.check(lib.lce_hip_memcpy_d2h(h.ctypes.data_as(C.c_void_p), C.c_void_p(d), C.c_size_t(h.nbytes), None))
        amd.check(lib.lce_hip_stream_synchronize(None))
        for t, h in zip(sec.outputs, host):
            assert np.array_equal(h.view(np.uint8), want[t].view(np.uint8)), (n, t)
        for d in ins + outs:
            amd.check(lib.lce_hip_free(C.c_void_p(d)))
    assert model.run_stats()[1] == 1                      # the float convolution + its LceQuantize: one launch
    with pytest.raises(amd.LceHipError):
        model.run_section(5, 1, [0], [0])
    with pytest.raises(amd.LceHipError):
        model.run_section(0, 1, [0] * len(sec.inputs), outs)     # null input pointer


def test_run_section_replays_a_recorded_hip_graph():
    """lce_tflite_model_use_hip_graphs: on a stream of its own the section runs eagerly once, is recorded at the second call and
    replayed afterwards -- same bytes as the eager run and as the oracle every time, new inputs in the same buffers included; other
    tensor pointers get their own recording; a larger batch (which reallocates the model's intermediates) drops the recordings; the
    null stream is never recorded.  No torch: lce_hip_* only."""
    import ctypes as C
    data, p = small_model(24)
    model = mr.LceModel(data)
    sec = model.sections[0]
    lib = amd.lib()
    stream = C.c_void_p()
    amd.check(lib.lce_hip_stream_create(C.byref(stream)))

    def buffers(n):
        ins, outs, host = [], [], []
        for t in sec.inputs:
            _, nbytes = model.section_tensor_shape(0, t, n)
            d = C.c_void_p()
            amd.check(lib.lce_hip_malloc(C.byref(d), C.c_size_t(nbytes)))
            ins.append(d.value)
        for t in sec.outputs:
            dims, nbytes = model.section_tensor_shape(0, t, n)
            d = C.c_void_p()
            amd.check(lib.lce_hip_malloc(C.byref(d), C.c_size_t(nbytes)))
            outs.append(d.value)
            host.append(np.empty(dims, mr._NP[model.tensors[t].type]))
        return ins, outs, host

    def run_and_check(n, seed, ins, outs, host, st):
        x = synth.rng(seed).uniform(-1.5, 1.5, (n, 12, 12, 64)).astype(np.float32)
        want = dict(zip(model.outputs, oracle_forward(x, p)))
        amd.check(lib.lce_hip_memcpy_h2d(C.c_void_p(ins[0]), x.ctypes.data_as(C.c_void_p), C.c_size_t(x.nbytes), st))
        for d, h in zip(outs, host):
            amd.check(lib.lce_hip_memset(C.c_void_p(d), 0x5A, C.c_size_t(h.nbytes), st))
        model.run_section(0, n, ins, outs, stream=st.value or 0)
        for d, h in zip(outs, host):
            amd.check(lib.lce_hip_memcpy_d2h(h.ctypes.data_as(C.c_void_p), C.c_void_p(d), C.c_size_t(h.nbytes), st))
        amd.check(lib.lce_hip_stream_synchronize(st))
        for t, h in zip(sec.outputs, host):
            assert np.array_equal(h.view(np.uint8), want[t].view(np.uint8)), (n, seed, t)

    model.use_hip_graphs(True)
    a = buffers(6)
    for k in range(4):                                     # eager, record + launch, replay, replay
        run_and_check(6, 50 + k, *a, stream)
        assert model.graph_stats() == [(0, 0), (1, 1), (1, 2), (1, 3)][k], k
    assert model.run_stats()[1] == 1                       # (the fused LceQuantize is counted for replays too)
    b = buffers(6)                                         # other tensors: their own recording
    for k in range(3):
        run_and_check(6, 60 + k, *b, stream)
    assert model.graph_stats() == (2, 5)
    run_and_check(6, 70, *a, stream)                       # the first recording is still good
    assert model.graph_stats() == (2, 6)
    run_and_check(6, 71, *a, C.c_void_p(None))             # the null stream: eager, nothing recorded
    assert model.graph_stats() == (2, 6)
    c = buffers(9)                                         # larger batch: intermediates reallocated, recordings dropped
    for k in range(3):
        run_and_check(9, 80 + k, *c, stream)
    assert model.graph_stats() == (3, 8)
    for k in range(3):                                     # ... so the first key records again (eager, record, replay)
        run_and_check(6, 90 + k, *a, stream)
    assert model.graph_stats() == (4, 10)
    model.use_hip_graphs(False)
    run_and_check(6, 99, *a, stream)
    assert model.graph_stats() == (4, 10)
    for d in a[0] + a[1] + b[0] + b[1] + c[0] + c[1]:
        amd.check(lib.lce_hip_free(C.c_void_p(d)))
    amd.check(lib.lce_hip_stream_destroy(stream))
