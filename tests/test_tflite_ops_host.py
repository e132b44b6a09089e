"""CPU-side tests of the TFLite custom-op surface (Register_* + RegisterLCECustomOps):
option parsing, Prepare-time validation / shape inference / error messages -- everything
the reference's op glue does before a kernel runs (tflite/kernels/bconv2d.cc:85-300,
quantization.cc:19-74, bmaxpool.cc:20-77)."""
import json
import os

import numpy as np
import pytest

import flexbuf
import oracle_lib as O
import synth
import tflite_driver as T
from lce_amd import amd
from test_oracle_vs_float_conv import CASES, PADS, legal

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


@pytest.mark.parametrize("kat", ["bconv2d_custom_options", "bmaxpool_custom_options"])
def test_flexbuffer_known_answer(kat):
    """mlir/tests/legalize-lce.mlir:9,21 -- the exact bytes the converter emits."""
    buf = bytes.fromhex(KATS[kat]["flexbuffer_hex"])
    for key, want in KATS[kat]["expected"].items():
        assert T.flex_lookup(buf, key) == (False, want)
    assert T.flex_lookup(buf, "no_such_key") == (True, 0)
    # our writer reproduces the reference bytes, so written-then-read options are trustworthy
    if kat == "bconv2d_custom_options":
        assert flexbuf.bconv2d_options(3, 1, 1, 1, 1, 1, 0, 0) == buf


def test_flexbuffer_wide_values_and_garbage():
    buf = flexbuf.bconv2d_options(3072, 2, 3, 3, 2, 0, 1, 3)
    assert T.flex_lookup(buf, "channels_in") == (False, 3072)
    assert T.flex_lookup(buf, "fused_activation_function") == (False, 3)
    big = flexbuf.build_int_map({"channels_in": 70000, "x": -5})
    assert T.flex_lookup(big, "channels_in") == (False, 70000) and T.flex_lookup(big, "x") == (False, -5)
    for junk in (b"", b"\x00", b"\x01\x02\x03", bytes(range(40))):
        with pytest.raises(ValueError):
            T.flex_lookup(junk, "channels_in")


def test_flexbuffer_reader_survives_every_single_byte_corruption():
    """The options blob comes out of a file: every byte of a valid map replaced by 0x00 / 0x7f / 0xff (sizes
    and offsets that would wrap a pointer, widths of 8, counts of 2^32) must be refused or read as some
    value -- never crash.  Copies are placed at the END of an exactly-sized heap buffer so that a read past
    either end of the blob is a read out of the allocation (caught by the allocator's guard in debug runs;
    here the test asserts the call returns)."""
    for base in (flexbuf.bconv2d_options(64, 1, 1, 1, 1, 0, 1, 1), flexbuf.build_int_map({"channels_in": 70000, "x": -5})):
        for pos in range(len(base)):
            for v in (0x00, 0x7F, 0xFF, 0x08, 0x24):
                blob = bytearray(base)
                blob[pos] = v
                try:
                    T.flex_lookup(bytes(blob), "channels_in")
                except ValueError:
                    pass
        for cut in range(len(base)):
            try:
                T.flex_lookup(bytes(base[:cut]), "channels_in")
            except ValueError:
                pass


def _spec(case, sem):
    inp, flt, g, st, dil, pad, act = case
    padding, pv = PADS[pad]
    return O.ConvSpec(inp[0], inp[1], inp[2], inp[3], flt[0], flt[1], flt[2], g, st[0], st[1], dil[0],
                      dil[1], padding, pv, act, sem)


def test_bconv2d_prepare_shape_inference_on_reference_grid():
    checked = 0
    for case in CASES[::9]:
        for variant, sem in ((T.BCONV_REF, O.SEM_REFERENCE), (T.BCONV_OPT_INDIRECT, O.SEM_OPTIMIZED)):
            if not legal(case[0], case[1], case[2], case[5], sem):
                continue
            spec = _spec(case, sem)
            if spec.out_h <= 0 or spec.out_w <= 0:
                continue
            zero_pad = case[5] == "SAME"
            for dst in (O.DST_F32, O.DST_I8, O.DST_BITPACKED):
                if zero_pad and sem == O.SEM_OPTIMIZED and (dst != O.DST_F32 or spec.activation != O.ACT_NONE):
                    continue
                _, w, mul, bias = synth.conv_inputs(spec, 1)
                thr = O.thresholds_converter(spec, mul, bias)
                m, _, out = T.build_bconv2d(spec, dst, w, mul, bias, thr, variant, out_scale=0.5, out_zero_point=1)
                assert m.prepare() == 0, m.log
                assert m.shape(out) == spec.output_shape(dst)
                assert m.num_temporaries == 0       # implicit GEMM: no im2col temporary
                checked += 1
    assert checked > 150


def test_bconv2d_zero_padding_rules_and_messages():
    """tflite/kernels/bconv2d.cc:188-200; death tests tflite/tests/bconv2d_test.cc:858-917."""
    spec = O.ConvSpec(1, 16, 16, 64, 3, 3, 128, padding=O.PADDING_SAME, pad_values=0, activation=O.ACT_RELU)
    _, w, mul, bias = synth.conv_inputs(spec, 1)
    thr = np.zeros(128, np.int32)
    m, *_ = T.build_bconv2d(spec, O.DST_F32, w, mul, bias, thr, T.BCONV_OPT_BGEMM)
    assert m.prepare() == 1 and "Zero-padding is only supported by" in m.log
    spec.activation = O.ACT_NONE
    for dst in (O.DST_BITPACKED, O.DST_I8):
        m, *_ = T.build_bconv2d(spec, dst, w, mul, bias, thr, T.BCONV_OPT_BGEMM)
        assert m.prepare() == 1 and "Zero-padding is only supported by" in m.log
    m, *_ = T.build_bconv2d(spec, O.DST_F32, w, mul, bias, thr, T.BCONV_OPT_BGEMM)
    assert m.prepare() == 0, m.log
    for dst in (O.DST_F32, O.DST_BITPACKED, O.DST_I8):        # the reference kernel takes all three
        m, *_ = T.build_bconv2d(spec, dst, w, mul, bias, thr, T.BCONV_REF)
        assert m.prepare() == 0, m.log


def test_bconv2d_other_prepare_errors():
    spec = O.ConvSpec(1, 8, 8, 128, 3, 3, 64, groups=2)
    _, w, mul, bias = synth.conv_inputs(spec, 2)
    thr = np.zeros(64, np.int32)
    m, *_ = T.build_bconv2d(spec, O.DST_F32, w, mul, bias, thr, T.BCONV_OPT_BGEMM)
    assert m.prepare() == 1 and "Grouped binary convolutions are not supported with this kernel." in m.log
    for v in (T.BCONV_REF, T.BCONV_OPT_INDIRECT):
        m, *_ = T.build_bconv2d(spec, O.DST_F32, w, mul, bias, thr, v)
        assert m.prepare() == 0, m.log
    m, *_ = T.build_bconv2d(spec, O.DST_F32, w, mul, bias, thr, T.BCONV_OPT_INDIRECT, in_alloc=T.DYNAMIC)
    assert m.prepare() == 1 and "dynamic allocation type" in m.log
    # pad_values outside {0,1}: Init logs, Prepare fails (bconv2d.cc:109-112,143)
    one = O.ConvSpec(1, 8, 8, 64, 3, 3, 64)
    _, w1, mul1, bias1 = synth.conv_inputs(one, 3)
    bad = flexbuf.bconv2d_options(64, pad_values=2)
    m, *_ = T.build_bconv2d(one, O.DST_F32, w1, mul1, bias1, thr, options=bad)
    assert m.prepare() == 1 and "pad_values must be 0 or 1" in m.log
    # a missing attribute: Init cannot fail, Prepare must (bconv2d.cc:96-103,126-129,143)
    missing = flexbuf.build_int_map({"channels_in": 64, "stride_height": 1})
    m, *_ = T.build_bconv2d(one, O.DST_F32, w1, mul1, bias1, thr, options=missing)
    assert m.prepare() == 1 and "IsNull() was not true" in m.log
    # wrong threshold length for bitpacked output (:212-218)
    m, *_ = T.build_bconv2d(one, O.DST_BITPACKED, w1, mul1, bias1, np.zeros(64, np.int32))
    assert m.prepare() == 0, m.log


def test_register_lce_custom_ops_selection():
    """lce_ops_register.h:25-53: the two flags choose the LceBconv2d registration."""
    spec = O.ConvSpec(1, 8, 8, 64, 3, 3, 64, padding=O.PADDING_SAME, pad_values=0)
    _, w, mul, bias = synth.conv_inputs(spec, 4)
    thr = np.zeros(64, np.int32)
    # default (= OPT_BGEMM semantics): SAME-zero int8 is rejected; use_reference_bconv: accepted
    m, *_ = T.build_bconv2d(spec, O.DST_I8, w, mul, bias, thr, variant=0, use_resolver=True, out_scale=1.0)
    assert m.prepare() == 1
    m, *_ = T.build_bconv2d(spec, O.DST_I8, w, mul, bias, thr, variant=1, use_resolver=True, out_scale=1.0)
    assert m.prepare() == 0, m.log
    # use_indirect_bgemm accepts groups, the default does not
    g = O.ConvSpec(1, 8, 8, 128, 3, 3, 64, groups=2)
    _, wg, mulg, biasg = synth.conv_inputs(g, 5)
    m, *_ = T.build_bconv2d(g, O.DST_F32, wg, mulg, biasg, thr, variant=0, use_resolver=True)
    assert m.prepare() == 1
    m, *_ = T.build_bconv2d(g, O.DST_F32, wg, mulg, biasg, thr, variant=2, use_resolver=True)
    assert m.prepare() == 0, m.log
    for name in ("LceQuantize", "LceDequantize", "LceBMaxPool2d"):
        T.SingleOpModel(name, 0, use_resolver=True).close()
    with pytest.raises(ValueError):
        T.SingleOpModel("LceNotAnOp", 0, use_resolver=True)


@pytest.mark.parametrize("ttype", [T.FLOAT32, T.INT8, T.BOOL])
@pytest.mark.parametrize("channels", [1, 31, 32, 33, 68])
def test_quantize_dequantize_prepare(ttype, channels):
    q = T.SingleOpModel("LceQuantize")
    i = q.add_tensor(ttype, (1, 4, 4, channels))
    o = q.add_tensor(T.INT32, (0, 0, 0, 0))
    q.set_node([i], [o])
    assert q.prepare() == 0, q.log
    assert q.shape(o) == (1, 4, 4, (channels + 31) // 32)
    d = T.SingleOpModel("LceDequantize")
    i = d.add_tensor(T.INT32, (1, 4, 4, (channels + 31) // 32))
    o = d.add_tensor(ttype, (1, 4, 4, channels))
    d.set_node([i], [o])
    assert d.prepare() == 0, d.log
    bad = T.SingleOpModel("LceDequantize")
    i = bad.add_tensor(T.INT32, (1, 4, 4, (channels + 31) // 32 + 1))
    o = bad.add_tensor(ttype, (1, 4, 4, channels))
    bad.set_node([i], [o])
    assert bad.prepare() == 1
    wrong = T.SingleOpModel("LceQuantize")
    i = wrong.add_tensor(T.INT32, (1, 4))
    o = wrong.add_tensor(T.INT32, (1, 1))
    wrong.set_node([i], [o])
    assert wrong.prepare() == 1


def test_bmaxpool_prepare():
    m = T.SingleOpModel("LceBMaxPool2d")
    i = m.add_tensor(T.INT32, (2, 32, 32, 3))
    o = m.add_tensor(T.INT32, (0, 0, 0, 0))
    m.set_node([i], [o], bytes.fromhex(KATS["bmaxpool_custom_options"]["flexbuffer_hex"]))
    assert m.prepare() == 0, m.log
    assert m.shape(o) == (2, 16, 16, 3)       # mlir/tests/legalize-lce.mlir:20-21: 32x32 -> 16x16
    z = T.SingleOpModel("LceBMaxPool2d")
    i = z.add_tensor(T.INT32, (2, 8, 8, 3))
    o = z.add_tensor(T.INT32, (0, 0, 0, 0))
    z.set_node([i], [o], flexbuf.bmaxpool_options(2, 2, 0, 2, 0))
    assert z.prepare() == 1                    # stride 0 (bmaxpool.cc:52)


@pytest.mark.skipif(amd.device_count() > 0, reason="GPU-less container only")
def test_invoke_without_gpu_reports_error_instead_of_falling_back():
    spec = O.ConvSpec(1, 4, 4, 64, 3, 3, 16)
    x, w, mul, bias = synth.conv_inputs(spec, 1)
    m, xi, _ = T.build_bconv2d(spec, O.DST_F32, w, mul, bias, None)
    assert m.prepare() == 0
    m.set_data(xi, x)
    assert m.invoke() == 1 and "no CPU fallback" in m.log


def test_a_chain_of_ops_prepares_through_one_context():
    """The chain driver: LCE nodes behind one context whose GetExecutionPlan / GetNodeAndRegistration are forbidden to
    kernels, as TensorFlow Lite's are (the ops never call them: the host declares its graph, lce_ops_register.h); shape
    inference runs node by node; without a GPU invoke fails loudly at the first op."""
    m = T.ChainModel()
    x = m.add_tensor(T.FLOAT32, (2, 12, 10, 64))
    q = m.add_tensor(T.INT32, (0,) * 4)
    _, w1, mul1, bias1 = synth.conv_inputs(O.ConvSpec(2, 12, 10, 64, 3, 3, 96, padding=O.PADDING_SAME, pad_values=1), 1)
    f1 = m.add_tensor(T.INT32, w1.shape, w1, allocation=T.MMAP_RO)
    th = m.add_tensor(T.INT32, (96,), np.zeros(96, np.int32), allocation=T.MMAP_RO)
    c1 = m.add_tensor(T.INT32, (0,) * 4)
    p = m.add_tensor(T.INT32, (0,) * 4)
    m.add_node("LceQuantize", [x], [q])
    m.add_node("LceBconv2d", [q, f1, -1, -1, th], [c1], flexbuf.bconv2d_options(64, 1, 1, 1, 1, O.PADDING_SAME, 1, O.ACT_NONE))
    m.add_node("LceBMaxPool2d", [c1], [p], flexbuf.bmaxpool_options(2, 2, 2, 2, 1))
    m.declare_graph([p])
    assert m.prepare() == 0, m.log
    assert m.shape(q) == (2, 12, 10, 2) and m.shape(c1) == (2, 12, 10, 3) and m.shape(p) == (2, 6, 5, 3)
    if amd.device_count() == 0:
        assert m.invoke() == 1 and "no CPU fallback" in m.log
    assert "forbidden" not in m.log
    n0 = T.device_buffers()[0]
    m.close()                                   # a declared graph and its buffers go with the interpreter's nodes
    assert T.device_buffers()[0] <= n0
    with pytest.raises(ValueError):
        m.add_node("NoSuchOp", [x], [q])
