"""The .tflite reader and the plan-from-operator helper (include/lce_tflite_model.h, SURVEY.md 8(f)
row n3) on the CPU: field round trip against tests/tflite_writer.py, the reference's own
flexbuffer option bytes inside a model, plans built from a file equal plans built by hand, and
malformed buffers are refused instead of crashing.  No real converter output exists in this image;
see the header of csrc/tflite/tflite_flatbuffer_reader.h."""
import importlib
import json
import os

import numpy as np
import pytest

import flexbuf
import oracle_lib as O
import synth
from tflite_writer import ModelBuilder

amd = importlib.import_module("compute-engine_amd")
mr = importlib.import_module("compute-engine_amd.model_runner")
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def bconv_options(spec: O.ConvSpec) -> bytes:
    return flexbuf.bconv2d_options(channels_in=spec.channels_in, dilation_height_factor=spec.dilation_h,
                                   dilation_width_factor=spec.dilation_w, fused_activation_function=spec.activation,
                                   pad_values=spec.pad_values, padding=spec.padding, stride_height=spec.stride_h,
                                   stride_width=spec.stride_w)


def small_model(seed=0):
    """float in -> LceQuantize -> LceBconv2d(float) -> LceQuantize -> LceBconv2d(bitpacked, RELU)
    -> LceBMaxPool2d -> LceBconv2d(int8) ; second output: LceDequantize of the pooled bits."""
    H, C0 = 12, 64
    s1 = O.ConvSpec(1, H, H, C0, 3, 3, 96, padding=O.PADDING_SAME, pad_values=1)
    s2 = O.ConvSpec(1, H, H, 96, 3, 3, 40, 1, 2, 2, 1, 1, O.PADDING_VALID, 0, O.ACT_RELU)
    oh = s2.out_h
    s3 = O.ConvSpec(1, oh // 2, oh // 2, 40, 1, 1, 33)
    _, w1, m1, b1 = synth.conv_inputs(s1, seed + 1)
    _, w2, m2, b2 = synth.conv_inputs(s2, seed + 2)
    _, w3, m3, b3 = synth.conv_inputs(s3, seed + 3)
    thr2 = O.thresholds_converter(s2, m2, b2)
    sc3, zp3 = synth.int8_quant_params(seed + 3)
    b = ModelBuilder()
    t_in = b.tensor([1, H, H, C0], np.float32, "input")
    t_q1 = b.tensor([1, H, H, 2], np.int32, "q1")
    t_w1 = b.tensor(w1.shape, np.int32, "w1", w1)
    t_m1 = b.tensor([96], np.float32, "m1", m1)
    t_b1 = b.tensor([96], np.float32, "b1", b1)
    t_y1 = b.tensor([1, H, H, 96], np.float32, "y1")
    t_q2 = b.tensor([1, H, H, 3], np.int32, "q2")
    t_w2 = b.tensor(w2.shape, np.int32, "w2", w2)
    t_t2 = b.tensor([40], np.int32, "thr2", thr2)
    t_y2 = b.tensor([1, oh, oh, 2], np.int32, "y2")
    t_p = b.tensor([1, oh // 2, oh // 2, 2], np.int32, "pooled")
    t_w3 = b.tensor(w3.shape, np.int32, "w3", w3)
    t_m3 = b.tensor([33], np.float32, "m3", m3)
    t_b3 = b.tensor([33], np.float32, "b3", b3)
    t_y3 = b.tensor([1, oh // 2, oh // 2, 33], np.int8, "y3", scale=float(sc3), zero_point=zp3)
    t_d = b.tensor([1, oh // 2, oh // 2, 40], np.float32, "dequantized")
    b.inputs, b.outputs = [t_in], [t_y3, t_d]
    b.custom_op("LceQuantize", [t_in], [t_q1], b"")
    b.custom_op("LceBconv2d", [t_q1, t_w1, t_m1, t_b1, -1], [t_y1], bconv_options(s1))
    b.custom_op("LceQuantize", [t_y1], [t_q2], b"")
    b.custom_op("LceBconv2d", [t_q2, t_w2, -1, -1, t_t2], [t_y2], bconv_options(s2))
    b.custom_op("LceBMaxPool2d", [t_y2], [t_p], flexbuf.bmaxpool_options(2, 2, 2, 2, O.PADDING_VALID))
    b.custom_op("LceBconv2d", [t_p, t_w3, t_m3, t_b3, -1], [t_y3], bconv_options(s3))
    b.custom_op("LceDequantize", [t_p], [t_d], b"")
    params = dict(s1=s1, s2=s2, s3=s3, w=(w1, w2, w3), m=(m1, m2, m3), b=(b1, b2, b3), thr2=thr2, q3=(sc3, zp3))
    return b.finish(), params


def oracle_forward(x, p):
    n = x.shape[0]
    s1, s2, s3 = (s.with_batch(n) for s in (p["s1"], p["s2"], p["s3"]))
    y1 = O.bconv2d(s1, O.DST_F32, O.bitpack(x), p["w"][0], p["m"][0], p["b"][0])
    y2 = O.bconv2d(s2, O.DST_BITPACKED, O.bitpack(y1), p["w"][1], thresholds=p["thr2"])
    pooled = O.bmaxpool(y2, 2, 2, 2, 2, O.PADDING_VALID)
    y3 = O.bconv2d(s3, O.DST_I8, pooled, p["w"][2], p["m"][2], p["b"][2], out_scale=float(p["q3"][0]),
                   out_zero_point=p["q3"][1])
    return y3, O.unpack(pooled, 40, np.float32)


def test_reader_round_trips_every_field():
    data, p = small_model()
    assert data[4:8] == b"TFL3"
    m = mr.LceModel(data)
    assert [t.name for t in m.tensors][:3] == ["input", "q1", "w1"]
    assert m.tensors[0].shape == (1, 12, 12, 64) and m.tensors[0].type == mr.FLOAT32 and not m.tensors[0].constant
    assert m.tensors[2].constant and m.tensors[2].shape == p["w"][0].shape and m.tensors[2].type == mr.INT32
    y3 = m.tensors[m.outputs[0]]
    assert y3.type == mr.INT8 and y3.scale == pytest.approx(float(p["q3"][0])) and y3.zero_point == p["q3"][1]
    assert m.tensors[1].scale is None
    assert [o.custom_code for o in m.operators] == ["LceQuantize", "LceBconv2d", "LceQuantize", "LceBconv2d",
                                                    "LceBMaxPool2d", "LceBconv2d", "LceDequantize"]
    assert all(o.builtin_code == 32 for o in m.operators)
    assert m.operators[1].inputs[4] == -1 and m.operators[3].inputs[2:4] == [-1, -1]
    assert m.inputs == [0] and len(m.outputs) == 2
    o = m.operators[3]
    assert (o.option("stride_height"), o.option("padding"), o.option("fused_activation_function"),
            o.option("channels_in")) == (2, O.PADDING_VALID, O.ACT_RELU, 96)
    assert o.option("no_such_key") is None
    assert m.operators[4].option("filter_height") == 2


def test_reference_option_bytes_survive_inside_a_model():
    """The exact custom_options blob of mlir/tests/legalize-lce.mlir:9 as an operator's options."""
    k = KATS["bconv2d_custom_options"]
    blob = bytes.fromhex(k["flexbuffer_hex"])
    b = ModelBuilder()
    t0 = b.tensor([1, 4, 4, 1], np.int32, "x")
    t1 = b.tensor([8, 2, 2, 1], np.int32, "w", np.zeros((8, 2, 2, 1), np.int32))
    t2 = b.tensor([1, 3, 3, 8], np.float32, "y")
    b.inputs, b.outputs = [t0], [t2]
    b.custom_op("LceBconv2d", [t0, t1, -1, -1, -1], [t2], blob)
    m = mr.LceModel(b.finish())
    for key, want in k["expected"].items():
        assert m.operators[0].option(key) == want, key


def test_plan_from_model_equals_plan_built_by_hand():
    data, p = small_model(5)
    m = mr.LceModel(data)
    for op_index, spec, dst, kw in ((1, p["s1"], amd.F32, {}), (5, p["s3"], amd.I8,
                                    dict(out_scale=float(p["q3"][0]), out_zero_point=p["q3"][1]))):
        plan = m.bconv2d_plan(op_index, batch=7)
        assert plan.output_shape[0] == 7
        by_hand = amd.Bconv2dPlan(amd.ConvParams(7, spec.in_h, spec.in_w, spec.channels_in, spec.filter_h, spec.filter_w,
                                                 spec.channels_out, spec.groups, spec.stride_h, spec.stride_w,
                                                 spec.dilation_h, spec.dilation_w, spec.padding, spec.pad_values,
                                                 spec.activation, dst, amd.SEM_OPTIMIZED, **kw))
        i = {1: 0, 5: 2}[op_index]
        by_hand.set_weights(p["w"][i], p["m"][i], p["b"][i])
        assert plan.output_shape == by_hand.output_shape and plan.kernel_name() == by_hand.kernel_name()
        for a, b in zip(plan.folded(), by_hand.folded()):
            assert np.array_equal(a, b)
    with pytest.raises(amd.LceHipError, match="not an LceBconv2d"):
        m.bconv2d_plan(0, batch=1)


def test_prepare_errors_surface_through_the_model_path():
    """SAME padding with pad_values 0 and a fused RELU is refused by Prepare (bconv2d.cc:188-200)."""
    s = O.ConvSpec(1, 6, 6, 32, 3, 3, 8, padding=O.PADDING_SAME, pad_values=0, activation=O.ACT_RELU)
    _, w, mul, bias = synth.conv_inputs(s, 1)
    b = ModelBuilder()
    t0 = b.tensor([1, 6, 6, 1], np.int32, "x")
    t1 = b.tensor(w.shape, np.int32, "w", w)
    t2 = b.tensor([8], np.float32, "m", mul)
    t3 = b.tensor([8], np.float32, "b", bias)
    t4 = b.tensor([1, 6, 6, 8], np.float32, "y")
    b.inputs, b.outputs = [t0], [t4]
    b.custom_op("LceBconv2d", [t0, t1, t2, t3, -1], [t4], bconv_options(s))
    with pytest.raises(amd.LceHipError, match="Zero-padding is only supported by"):
        mr.LceModel(b.finish()).bconv2d_plan(0, batch=1)


def test_malformed_buffers_are_refused_not_crashed_on():
    data, _ = small_model(9)
    for bad in (b"", b"\0" * 7, b"\x08\0\0\0XXXX" + data[8:], data[:8]):
        with pytest.raises(ValueError, match="not a readable TFLite model"):
            mr.LceModel(bad)
    g = synth.rng(4)
    opened = 0
    for cut in list(range(8, len(data), 97)) + [len(data) - 1]:
        try:
            mr.LceModel(data[:cut])
            opened += 1
        except ValueError:
            pass
    # random corruption: any outcome but a crash; a model that still opens must stay walkable
    for _ in range(300):
        d = bytearray(data)
        for pos in g.integers(8, len(d), 6):
            d[pos] = int(g.integers(0, 256))
        try:
            m = mr.LceModel(bytes(d))
        except ValueError:
            continue
        for i, op in enumerate(m.operators):
            op.option("padding")
            if op.custom_code == "LceBconv2d":
                try:
                    m.bconv2d_plan(i, batch=1)
                except amd.LceHipError:
                    pass


def test_interpreter_refuses_to_predict_graphs_with_builtin_ops():
    b = ModelBuilder()
    t0 = b.tensor([1, 4, 4, 32], np.float32, "x")
    t1 = b.tensor([1, 4, 4, 32], np.float32, "y")
    b.inputs, b.outputs = [t0], [t1]
    b.builtin_op(19, [t0], [t1])      # RELU
    it = mr.Interpreter(b.finish())
    assert not it.lce_only and it.sections == []
    with pytest.raises(NotImplementedError, match="only LCE custom ops"):
        it.predict(np.zeros((1, 4, 4, 32), np.float32))


CONV_2D, ADD, MAX_POOL_2D = 3, 0, 17      # schema.fbs BuiltinOperator values


def mixed_model(seed=0):
    """A QuickNet-shaped mixed graph (float stem, residual ADDs between binary convolutions, float pooling head):

        x --CONV_2D(builtin stem)--> s --LceQuantize--> q0 --LceBconv2d(float)--> y0 --ADD(s)--> r0
          r0 --LceQuantize--> q1 --LceBconv2d(float)--> y1 --ADD(r0)--> r1
          r1 --LceQuantize--> q2 --LceBconv2d(bitpacked)--> b2 --LceBMaxPool2d--> p2 --LceBconv2d(float)--> y3 --MAX_POOL_2D--> out
          b2 --LceDequantize--> d2 (second graph output)

    Binary sections: {Quantize, Bconv} (s -> y0), {Quantize, Bconv} (r0 -> y1), {Quantize, Bconv, BMaxPool, Bconv, Dequantize}
    (r1 -> y3, d2)."""
    H, C = 10, 64
    s_a = O.ConvSpec(1, H, H, C, 3, 3, C, padding=O.PADDING_SAME, pad_values=1)
    s_c = O.ConvSpec(1, H, H, C, 3, 3, 96, padding=O.PADDING_SAME, pad_values=1, activation=O.ACT_RELU)
    s_d = O.ConvSpec(1, H // 2, H // 2, 96, 3, 3, 32, padding=O.PADDING_SAME, pad_values=1)
    _, w0, m0, b0 = synth.conv_inputs(s_a, seed + 1)
    _, w1, m1, b1 = synth.conv_inputs(s_a, seed + 2)
    _, w2, m2, b2 = synth.conv_inputs(s_c, seed + 3)
    _, w3, m3, b3 = synth.conv_inputs(s_d, seed + 4)
    thr2 = O.thresholds_converter(s_c, m2, b2)
    b = ModelBuilder()
    f32 = lambda shape, name, data=None: b.tensor(shape, np.float32, name, data)
    i32 = lambda shape, name, data=None: b.tensor(shape, np.int32, name, data)
    x = f32([1, H, H, 3], "image")
    k = f32([C, 3, 3, 3], "stem_filter", synth.rng(seed).standard_normal((C, 3, 3, 3)).astype(np.float32))
    kb = f32([C], "stem_bias", np.zeros(C, np.float32))
    s = f32([1, H, H, C], "stem")
    q0, y0, r0 = i32([1, H, H, 2], "q0"), f32([1, H, H, C], "y0"), f32([1, H, H, C], "r0")
    q1, y1, r1 = i32([1, H, H, 2], "q1"), f32([1, H, H, C], "y1"), f32([1, H, H, C], "r1")
    q2, bb2, p2 = i32([1, H, H, 2], "q2"), i32([1, H, H, 3], "b2"), i32([1, H // 2, H // 2, 3], "p2")
    y3, out, d2 = f32([1, H // 2, H // 2, 32], "y3"), f32([1, 2, 2, 32], "pooled"), f32([1, H, H, 96], "d2")
    tw = [i32(w.shape, "w%d" % i, w) for i, w in enumerate((w0, w1, w2, w3))]
    tm = [f32([len(m)], "m%d" % i, m) for i, m in enumerate((m0, m1, m2, m3))]
    tb = [f32([len(v)], "b%d" % i, v) for i, v in enumerate((b0, b1, b2, b3))]
    tthr = i32([96], "thr2", thr2)
    b.inputs, b.outputs = [x], [out, d2]
    b.builtin_op(CONV_2D, [x, k, kb], [s])                                             # 0
    b.custom_op("LceQuantize", [s], [q0], b"")                                         # 1
    b.custom_op("LceBconv2d", [q0, tw[0], tm[0], tb[0], -1], [y0], bconv_options(s_a))  # 2
    b.builtin_op(ADD, [y0, s], [r0])                                                   # 3
    b.custom_op("LceQuantize", [r0], [q1], b"")                                        # 4
    b.custom_op("LceBconv2d", [q1, tw[1], tm[1], tb[1], -1], [y1], bconv_options(s_a))  # 5
    b.builtin_op(ADD, [y1, r0], [r1])                                                  # 6
    b.custom_op("LceQuantize", [r1], [q2], b"")                                        # 7
    b.custom_op("LceBconv2d", [q2, tw[2], -1, -1, tthr], [bb2], bconv_options(s_c))     # 8
    b.custom_op("LceBMaxPool2d", [bb2], [p2], flexbuf.bmaxpool_options(2, 2, 2, 2, O.PADDING_VALID))   # 9
    b.custom_op("LceBconv2d", [p2, tw[3], tm[3], tb[3], -1], [y3], bconv_options(s_d))  # 10
    b.builtin_op(MAX_POOL_2D, [y3], [out])                                             # 11
    b.custom_op("LceDequantize", [bb2], [d2], b"")                                     # 12
    ids = dict(s=s, y0=y0, r0=r0, y1=y1, r1=r1, y3=y3, d2=d2, b2=bb2)
    params = dict(specs=(s_a, s_a, s_c, s_d), w=(w0, w1, w2, w3), m=(m0, m1, m2, m3), b=(b0, b1, b2, b3), thr2=thr2)
    return b.finish(), ids, params


def test_binary_sections_of_a_mixed_graph():
    """The partition of include/lce_tflite_model.h on a QuickNet-shaped graph: three sections, their boundary tensors, and --
    the LceDequantize that comes LAST in the file belongs to the third section (it can run as soon as b2 exists)."""
    data, t, _ = mixed_model()
    it = mr.Interpreter(data)
    assert not it.lce_only
    secs = it.sections
    assert [s.ops for s in secs] == [[1, 2], [4, 5], [7, 8, 9, 10, 12]]
    assert [s.inputs for s in secs] == [[t["s"]], [t["r0"]], [t["r1"]]]
    assert [s.outputs for s in secs] == [[t["y0"]], [t["y1"]], sorted([t["y3"], t["d2"]])]
    with pytest.raises(NotImplementedError, match="run_section"):
        it.predict(np.zeros((1, 10, 10, 3), np.float32))
    # an LCE-only graph is ONE section that reads the graph inputs and delivers the graph outputs
    data, _ = small_model(3)
    one = mr.Interpreter(data)
    assert one.lce_only and len(one.sections) == 1
    assert one.sections[0].ops == list(range(7)) and one.sections[0].inputs == one.model.inputs
    assert sorted(one.sections[0].outputs) == sorted(one.model.outputs)


def test_model_abi_exports_every_declared_symbol():
    """include/lce_tflite_model.h <-> liblce_tflite_ops.so."""
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "lce_tflite_model.h")).read()
    names = sorted(set(re.findall(r"\b(lce_tflite_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 11
    lib = mr.tflite_lib()
    for n in names:
        assert hasattr(lib, n), n


def test_section_partition_terminates_on_a_cyclic_graph():
    """A malformed file whose two LCE ops feed each other: no operator ever becomes ready, the partition gives up after two
    idle epochs instead of spinning, and there is no section to run."""
    b = ModelBuilder()
    t0 = b.tensor([1, 4, 4, 2], np.int32, "a")
    t1 = b.tensor([1, 4, 4, 2], np.int32, "b")
    x = b.tensor([1, 4, 4, 64], np.float32, "x")
    b.inputs, b.outputs = [x], [t1]
    b.custom_op("LceBMaxPool2d", [t1], [t0], flexbuf.bmaxpool_options(1, 1, 1, 1, O.PADDING_VALID))
    b.custom_op("LceBMaxPool2d", [t0], [t1], flexbuf.bmaxpool_options(1, 1, 1, 1, O.PADDING_VALID))
    m = mr.LceModel(b.finish())
    assert m.sections == []


def test_section_shapes_at_any_batch_without_a_device_and_run_fails_loudly_without_one():
    """lce_tflite_model_section_tensor_shape is the ops' own shape inference (quantization.cc:19-41, bmaxpool.cc:41-77,
    bconv2d.cc:137-300) at a caller-chosen batch, host-only; lce_tflite_model_run_section has no CPU fallback."""
    data, p = small_model(40)
    m = mr.LceModel(data)
    sec = m.sections[0]
    assert sec.inputs == m.inputs and sorted(sec.outputs) == sorted(m.outputs)
    oh = p["s2"].out_h
    by_name = {t.name: i for i, t in enumerate(m.tensors)}
    for n in (1, 7):
        assert m.section_tensor_shape(0, by_name["input"], n) == ((n, 12, 12, 64), n * 12 * 12 * 64 * 4)
        assert m.section_tensor_shape(0, by_name["q1"], n) == ((n, 12, 12, 2), n * 12 * 12 * 2 * 4)
        assert m.section_tensor_shape(0, by_name["y1"], n) == ((n, 12, 12, 96), n * 12 * 12 * 96 * 4)
        assert m.section_tensor_shape(0, by_name["q2"], n)[0] == (n, 12, 12, 3)
        assert m.section_tensor_shape(0, by_name["y2"], n)[0] == (n, oh, oh, 2)
        assert m.section_tensor_shape(0, by_name["pooled"], n)[0] == (n, oh // 2, oh // 2, 2)
        assert m.section_tensor_shape(0, by_name["y3"], n) == ((n, oh // 2, oh // 2, 33), n * (oh // 2) ** 2 * 33)
        assert m.section_tensor_shape(0, by_name["dequantized"], n)[0] == (n, oh // 2, oh // 2, 40)
    assert m.run_stats()[0] == 6                       # three convolutions x two batch sizes, planned on the host
    with pytest.raises(amd.LceHipError):
        m.section_tensor_shape(0, by_name["w1"], 1)    # a constant: not a tensor the section computes or is fed
    with pytest.raises(amd.LceHipError):
        m.section_tensor_shape(3, 0, 1)
    if amd.device_count() == 0:
        with pytest.raises(amd.LceHipError, match="no CPU fallback"):
            m.run_section(0, 1, [8], [8, 8])


def test_a_convolution_whose_declared_input_shape_disagrees_with_its_producer_is_refused():
    """Round 4's advisor: the plan of an LceBconv2d is built from the FILE's shape of its input tensor while the buffer it reads is
    sized from the walk's own shape inference; a file in which the two disagree (hand-edited, dynamic dimensions, an int8 tensor
    wired into a convolution) must be refused by the walk -- not run a convolution past the end of a scratch buffer."""
    spec = O.ConvSpec(1, 8, 8, 64, 3, 3, 64, padding=O.PADDING_SAME, pad_values=1)
    _, w, mul, bias = synth.conv_inputs(spec, 3)

    def build(q_shape, q_dtype=np.int32, pool=False):
        b = ModelBuilder()
        x = b.tensor([1, 8, 8, 64], np.float32, "x")
        q = b.tensor(q_shape, q_dtype, "q")
        y = b.tensor([1, 8, 8, 64], np.float32, "y")
        tw, tm, tb = b.tensor(w.shape, np.int32, "w", w), b.tensor([64], np.float32, "m", mul), b.tensor([64], np.float32, "b", bias)
        b.inputs, b.outputs = [x], [y]
        b.custom_op("LceQuantize", [x], [q], b"")
        b.custom_op("LceBconv2d", [q, tw, tm, tb, -1], [y], bconv_options(spec))
        return b.finish(), y

    good, y = build([1, 8, 8, 2])
    assert mr.LceModel(good).section_tensor_shape(0, y, 5)[0] == (5, 8, 8, 64)
    # the quantize produces 8x8x2 words from the 8x8x64 input; the file says the convolution reads 16x16x2
    bad, y = build([1, 16, 16, 2])
    with pytest.raises(amd.LceHipError, match="declared shape does not match"):
        mr.LceModel(bad).section_tensor_shape(0, y, 5)


def test_partition_starts_with_the_kind_of_the_first_ready_operator():
    """graph_info.cc takes the first epoch's kind from the first ready node in execution order.  Here a builtin stem op (index 0)
    and an LceQuantize of ANOTHER graph input (index 1) are both ready at the start: the builtin epoch runs first, so the
    quantize shares a section with the LceBconv2d behind it instead of getting a leading section of its own."""
    spec = O.ConvSpec(1, 8, 8, 64, 3, 3, 64, padding=O.PADDING_SAME, pad_values=1)
    _, w, mul, bias = synth.conv_inputs(spec, 4)
    b = ModelBuilder()
    x0 = b.tensor([1, 8, 8, 64], np.float32, "x0")
    x1 = b.tensor([1, 8, 8, 64], np.float32, "x1")
    r = b.tensor([1, 8, 8, 64], np.float32, "relu")
    q1 = b.tensor([1, 8, 8, 2], np.int32, "q1")
    qr = b.tensor([1, 8, 8, 2], np.int32, "qr")
    y = b.tensor([1, 8, 8, 64], np.float32, "y")
    tw, tm, tb = b.tensor(w.shape, np.int32, "w", w), b.tensor([64], np.float32, "m", mul), b.tensor([64], np.float32, "b", bias)
    b.inputs, b.outputs = [x0, x1], [y, q1]
    b.builtin_op(19, [x0], [r])                         # 0: RELU (builtin), ready at the start
    b.custom_op("LceQuantize", [x1], [q1], b"")         # 1: LCE, ready at the start too
    b.custom_op("LceQuantize", [r], [qr], b"")          # 2
    b.custom_op("LceBconv2d", [qr, tw, tm, tb, -1], [y], bconv_options(spec))   # 3
    m = mr.LceModel(b.finish())
    assert [s.ops for s in m.sections] == [[1, 2, 3]]


def test_partition_is_linear_in_the_model_size():
    """Model open on an adversarially long chain: 4000 operators partition in well under a second (round 3's partition was
    quadratic in the operator count)."""
    import time
    b = ModelBuilder()
    prev = b.tensor([1, 4, 4, 2], np.int32, "t0")
    b.inputs = [prev]
    for i in range(4000):
        nxt = b.tensor([1, 4, 4, 2], np.int32, "t%d" % (i + 1))
        b.custom_op("LceBMaxPool2d", [prev], [nxt], flexbuf.bmaxpool_options(1, 1, 1, 1, O.PADDING_VALID))
        prev = nxt
    b.outputs = [prev]
    data = b.finish()
    t = time.perf_counter()
    m = mr.LceModel(data)
    dt = time.perf_counter() - t
    assert len(m.sections) == 1 and len(m.sections[0].ops) == 4000 and m.sections[0].outputs == [4000]
    assert dt < 5.0, dt
