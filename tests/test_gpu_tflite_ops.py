"""GPU end-to-end tests of the TFLite custom-op surface: Init -> Prepare -> Invoke through
Register_BCONV_2D{,_REF,_OPT_BGEMM,_OPT_INDIRECT_BGEMM}, Register_QUANTIZE,
Register_DEQUANTIZE, Register_BMAXPOOL_2D with host (interpreter-arena) tensors, checked
against the oracle.  This is the shape of the reference's own op tests
(tflite/tests/bconv2d_test.cc, quantization_test.cc, bmaxpool_test.cc)."""
import zlib

import numpy as np
import pytest

import flexbuf
import oracle_lib as O
import synth
import tflite_driver as T
from test_oracle_vs_float_conv import CASES, PADS, _id, legal

pytestmark = pytest.mark.gpu

VARIANTS = {T.BCONV_REF: O.SEM_REFERENCE, T.BCONV_OPT_BGEMM: O.SEM_OPTIMIZED,
            T.BCONV_OPT_INDIRECT: O.SEM_OPTIMIZED, T.BCONV_DEFAULT: O.SEM_OPTIMIZED}


@pytest.mark.parametrize("case", CASES[:72:3] + CASES[72::23], ids=_id)
def test_bconv2d_op_all_registrations(case):
    inp, flt, g, st, dil, pad, act = case
    padding, pv = PADS[pad]
    for variant, sem in VARIANTS.items():
        if not legal(inp, flt, g, pad, sem):
            continue
        if g > 1 and variant in (T.BCONV_OPT_BGEMM, T.BCONV_DEFAULT):
            continue          # rejected by Prepare (bconv2d.cc:173-175), tested on the host side
        spec = O.ConvSpec(inp[0], inp[1], inp[2], inp[3], flt[0], flt[1], flt[2], g, st[0], st[1], dil[0],
                          dil[1], padding, pv, act, sem)
        if spec.out_h <= 0 or spec.out_w <= 0:
            continue
        seed = zlib.crc32(_id(case).encode()) & 0xFFFF
        x, w, mul, bias = synth.conv_inputs(spec, seed)
        thr = O.thresholds_optest(spec, mul, bias)
        scale, zp = synth.int8_quant_params(seed)
        zero_pad = pad == "SAME"
        for dst in (O.DST_F32, O.DST_I8, O.DST_BITPACKED):
            if zero_pad and sem == O.SEM_OPTIMIZED and (dst != O.DST_F32 or act != O.ACT_NONE):
                continue
            m, xi, oi = T.build_bconv2d(spec, dst, w, mul, bias, thr, variant, out_scale=float(scale), out_zero_point=zp)
            assert m.prepare() == 0, m.log
            m.set_data(xi, x)
            assert m.invoke() == 0, m.log
            want = O.bconv2d(spec, dst, x, w, mul, bias, thresholds=thr, out_scale=float(scale), out_zero_point=zp)
            assert np.array_equal(m.get(oi).view(np.uint8), want.view(np.uint8)), (variant, dst)
            # second invoke with new input: constants are folded once, results still right
            x2 = synth.random_words(synth.rng(seed + 1), spec.input_shape(), spec.channels_in)
            m.set_data(xi, x2)
            assert m.invoke() == 0, m.log
            want2 = O.bconv2d(spec, dst, x2, w, mul, bias, thresholds=thr, out_scale=float(scale), out_zero_point=zp)
            assert np.array_equal(m.get(oi).view(np.uint8), want2.view(np.uint8))


@pytest.mark.parametrize("ttype,dtype", [(T.FLOAT32, np.float32), (T.INT8, np.int8), (T.BOOL, np.bool_)])
@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (1, 4, 4, 1), (1, 4, 4, 2), (1, 4, 4, 31), (1, 4, 4, 32),
                                   (1, 4, 4, 33), (1, 4, 4, 64), (1, 4, 4, 68)])
def test_quantize_dequantize_ops_round_trip(ttype, dtype, shape):
    """tflite/tests/quantization_test.cc:75-130."""
    g = synth.rng(sum(shape))
    signs = np.where(g.random(shape) < 0.5, -1.0, 1.0).astype(np.float32)
    n, zp = int(g.integers(1, 21)), int(g.integers(-20, 21))
    scale = float(np.float32(1.0) / np.float32(n))
    data = {np.float32: signs, np.int8: (zp + n * signs).astype(np.int8), np.bool_: signs > 0}[dtype]
    q = T.SingleOpModel("LceQuantize")
    qi = q.add_tensor(ttype, shape, data, scale=scale if dtype == np.int8 else 0.0, zero_point=zp if dtype == np.int8 else 0)
    qo = q.add_tensor(T.INT32, (0,) * 4)
    q.set_node([qi], [qo])
    assert q.prepare() == 0 and q.invoke() == 0, q.log
    packed = q.get(qo)
    assert np.array_equal(packed, O.bitpack(data, zp if dtype == np.int8 else 0))
    d = T.SingleOpModel("LceDequantize")
    di = d.add_tensor(T.INT32, packed.shape, packed)
    do = d.add_tensor(ttype, shape, scale=scale if dtype == np.int8 else 0.0, zero_point=zp if dtype == np.int8 else 0)
    d.set_node([di], [do])
    assert d.prepare() == 0 and d.invoke() == 0, d.log
    assert np.array_equal(d.get(do), data)


def test_bmaxpool_op():
    x = synth.random_words(synth.rng(7), (2, 9, 11, 3))
    for fh, fw, sh, sw, pad in ((2, 2, 2, 2, 0), (3, 3, 2, 2, 0), (3, 2, 1, 2, 1)):
        m = T.SingleOpModel("LceBMaxPool2d")
        i = m.add_tensor(T.INT32, x.shape, x)
        o = m.add_tensor(T.INT32, (0,) * 4)
        m.set_node([i], [o], flexbuf.bmaxpool_options(fh, fw, sh, sw, pad))
        assert m.prepare() == 0 and m.invoke() == 0, m.log
        assert np.array_equal(m.get(o), O.bmaxpool(x, fh, fw, sh, sw, pad))


def test_prepare_again_after_resize_redoes_setup():
    """bconv2d.cc:295-297."""
    spec = O.ConvSpec(2, 6, 6, 64, 3, 3, 32, padding=O.PADDING_SAME, pad_values=1)
    x, w, mul, bias = synth.conv_inputs(spec, 9)
    m, xi, oi = T.build_bconv2d(spec, O.DST_F32, w, mul, bias, None)
    for _ in range(2):
        assert m.prepare() == 0, m.log
        m.set_data(xi, x)
        assert m.invoke() == 0, m.log
        assert np.array_equal(m.get(oi), O.bconv2d(spec, O.DST_F32, x, w, mul, bias))


# ---------------------------------------------------------------------------------------------------------------------
# device residency between LCE ops (csrc/tflite/lce_ops.cc, namespace resident)
def _binary_section(m, b=3, h=12, w_=10, c0=64, c1=96, c2=64, host_reader=False, seed=5):
    """LceQuantize -> LceBconv2d (bitpacked out) -> LceBMaxPool2d 2x2 -> LceBconv2d (float out), as nodes of `m`.
    Returns (input index, output index, index of the first conv's output, reference fn)."""
    g = synth.rng(seed)
    s1 = O.ConvSpec(b, h, w_, c0, 3, 3, c1, padding=O.PADDING_SAME, pad_values=1)
    s2 = O.ConvSpec(b, h // 2, w_ // 2, c1, 3, 3, c2, padding=O.PADDING_SAME, pad_values=1, activation=O.ACT_RELU)
    _, w1, mul1, bias1 = synth.conv_inputs(s1, seed + 1)
    _, w2, mul2, bias2 = synth.conv_inputs(s2, seed + 2)
    thr1 = O.thresholds_converter(s1, mul1, bias1)
    x = g.standard_normal((b, h, w_, c0)).astype(np.float32)
    t_x = m.add_tensor(T.FLOAT32, x.shape, x)
    t_q = m.add_tensor(T.INT32, (0,) * 4)
    t_w1 = m.add_tensor(T.INT32, s1.filter_shape(), w1, allocation=T.MMAP_RO)
    t_thr = m.add_tensor(T.INT32, (c1,), thr1, allocation=T.MMAP_RO)
    t_c1 = m.add_tensor(T.INT32, (0,) * 4)
    t_p = m.add_tensor(T.INT32, (0,) * 4)
    t_w2 = m.add_tensor(T.INT32, s2.filter_shape(), w2, allocation=T.MMAP_RO)
    t_m2 = m.add_tensor(T.FLOAT32, (c2,), mul2, allocation=T.MMAP_RO)
    t_b2 = m.add_tensor(T.FLOAT32, (c2,), bias2, allocation=T.MMAP_RO)
    t_y = m.add_tensor(T.FLOAT32, (0,) * 4)
    m.add_node("LceQuantize", [t_x], [t_q])
    m.add_node("LceBconv2d", [t_q, t_w1, -1, -1, t_thr], [t_c1], flexbuf.bconv2d_options(c0, 1, 1, 1, 1, s1.padding, 1, O.ACT_NONE))
    t_copy = None
    if host_reader:                                   # a CPU kernel that reads the first conv's output from the arena
        t_copy = m.add_tensor(T.INT32, (0,) * 4)
        m.add_node("HostCopy", [t_c1], [t_copy])
    m.add_node("LceBMaxPool2d", [t_c1], [t_p], flexbuf.bmaxpool_options(2, 2, 2, 2, 1))
    m.add_node("LceBconv2d", [t_p, t_w2, t_m2, t_b2, -1], [t_y], flexbuf.bconv2d_options(c1, 1, 1, 1, 1, s2.padding, 1, O.ACT_RELU))

    def reference(xin):
        q = O.bitpack(xin)
        c = O.bconv2d(s1, O.DST_BITPACKED, q, w1, thresholds=thr1)
        p = O.bmaxpool(c, 2, 2, 2, 2, 1)
        return c, O.bconv2d(s2, O.DST_F32, p, w2, mul2, bias2)
    return t_x, t_y, t_c1, t_copy, x, reference


def test_a_binary_section_crosses_pcie_once_in_each_direction():
    """LceQuantize -> LceBconv2d -> LceBMaxPool2d -> LceBconv2d through RegisterLCECustomOps: tensors that only LCE ops read
    stay in HBM (the reference hands them over in the arena, bconv2d.cc:550-564, quantization.cc:76-114, bmaxpool.cc:79-91),
    so, once the host has declared its graph (lce_ops_register.h), one invoke makes exactly ONE upload (the float input)
    and ONE download (the float output), bit-equal to the oracle -- also for a second invoke with new input data."""
    m = T.ChainModel()
    t_x, t_y, t_c1, _, x, reference = _binary_section(m)
    T.set_residency(True)
    m.declare_graph([t_y])
    assert m.prepare() == 0, m.log
    for k in range(2):
        xin = x if k == 0 else (x[::-1] * -1.0).copy()
        m.set_data(t_x, xin)
        T.transfer_counts(reset=True)
        assert m.invoke() == 0, m.log
        up, down, up_bytes, down_bytes = T.transfer_counts()
        _, want = reference(xin)
        got = m.get(t_y)
        assert np.array_equal(got.view(np.int32), want.view(np.int32))
        assert (up, down) == (1, 1), (up, down)
        assert up_bytes == xin.nbytes and down_bytes == want.nbytes
    assert "forbidden" not in m.log                      # the ops never call the delegate-only context functions


def test_without_a_declared_graph_every_output_is_in_the_arena():
    """The default (and every caller of the reference): no declaration -> each op copies its output back, as the reference's
    kernels leave theirs in the arena (bconv2d.cc:550-564).  Nothing can be stale; the ops ask the context nothing.
    Declaring the graph AFTER AllocateTensors switches residency on from the next invoke; forgetting it switches it off."""
    m = T.ChainModel()
    t_x, t_y, t_c1, _, x, reference = _binary_section(m)
    T.set_residency(True)
    assert m.prepare() == 0, m.log
    m.set_data(t_x, x)
    c1, want = reference(x)
    for declared, expect in ((False, (4, 4)), (True, (1, 1)), (False, (4, 4))):
        if declared:
            m.declare_graph([t_y])
        else:
            m.forget_graph()
        m.set_data(t_c1, np.full(m.shape(t_c1), 0x5A5A5A5A, np.int32) if m.shape(t_c1)[0] else np.zeros(m.shape(t_c1), np.int32))
        T.transfer_counts(reset=True)
        assert m.invoke() == 0, m.log
        up, down, _, _ = T.transfer_counts()
        assert np.array_equal(m.get(t_y).view(np.int32), want.view(np.int32))
        if not declared:
            assert np.array_equal(m.get(t_c1), c1)                 # the arena copy is current
        assert (up, down) == expect, (declared, up, down)
    assert "forbidden" not in m.log


def test_a_graph_output_that_also_feeds_an_lce_op_is_copied_back():
    """The case the round-3 layer got wrong: the first convolution's bitpacked output is BOTH a graph output and the
    pooling op's input.  Declared as an output it is downloaded (one more copy) and the LCE reader still takes the device
    buffer (no extra upload)."""
    m = T.ChainModel()
    t_x, t_y, t_c1, _, x, reference = _binary_section(m)
    T.set_residency(True)
    m.declare_graph([t_y, t_c1])
    assert m.prepare() == 0, m.log
    m.set_data(t_x, x)
    m.set_data(t_c1, np.full(m.shape(t_c1), 0x5A5A5A5A, np.int32))          # poison the arena copy
    T.transfer_counts(reset=True)
    assert m.invoke() == 0, m.log
    up, down, _, _ = T.transfer_counts()
    c1, want = reference(x)
    assert np.array_equal(m.get(t_c1), c1)
    assert np.array_equal(m.get(t_y).view(np.int32), want.view(np.int32))
    assert (up, down) == (1, 2), (up, down)


def test_device_buffers_are_bounded_and_die_with_the_interpreter():
    """One device buffer per tensor an LCE op produces + ONE shared staging buffer per context for inputs no LCE op
    produced (round 3 kept one per such tensor, forever); all of them are released with the interpreter's nodes."""
    n0, _ = T.device_buffers()
    m = T.ChainModel()
    t_x, t_y, t_c1, _, x, reference = _binary_section(m)
    T.set_residency(False)                               # every op stages its input: the worst case for staging buffers
    try:
        assert m.prepare() == 0, m.log
        m.set_data(t_x, x)
        for _ in range(3):
            assert m.invoke() == 0, m.log
        n1, bytes1 = T.device_buffers()
        # LceQuantize and LceBMaxPool2d each hold their output's buffer and share the staging buffer (the convolutions ran
        # host to host inside lce_hip_bconv2d_run_host)
        assert n1 - n0 == 3, (n0, n1)
        assert bytes1 <= 4 * x.nbytes
    finally:
        T.set_residency(True)
    m.close()
    assert T.device_buffers()[0] == n0


def test_a_tensor_with_a_host_reader_is_still_copied_back():
    m = T.ChainModel()
    t_x, t_y, t_c1, t_copy, x, reference = _binary_section(m, host_reader=True)
    T.set_residency(True)
    m.declare_graph([t_y, t_copy])
    assert m.prepare() == 0, m.log
    m.set_data(t_x, x)
    T.transfer_counts(reset=True)
    assert m.invoke() == 0, m.log
    up, down, _, _ = T.transfer_counts()
    c1, want = reference(x)
    assert np.array_equal(m.get(t_copy), c1) and np.array_equal(m.get(t_c1), c1)       # the arena copy is current
    assert np.array_equal(m.get(t_y).view(np.int32), want.view(np.int32))
    assert (up, down) == (1, 2), (up, down)                                             # ... and it cost one more download


def test_residency_can_be_switched_off():
    m = T.ChainModel()
    t_x, t_y, t_c1, _, x, reference = _binary_section(m)
    T.set_residency(False)
    m.declare_graph([t_y])                               # (a declaration is ignored while residency is off)
    try:
        assert m.prepare() == 0, m.log
        m.set_data(t_x, x)
        T.transfer_counts(reset=True)
        assert m.invoke() == 0, m.log
        up, down, _, _ = T.transfer_counts()
        c1, want = reference(x)
        assert np.array_equal(m.get(t_y).view(np.int32), want.view(np.int32)) and np.array_equal(m.get(t_c1), c1)
        # every op stages its own tensors again: LceQuantize and LceBMaxPool2d through the staging copies, the two
        # convolutions inside lce_hip_bconv2d_run_host (pipelined slices, counted as one pass each way)
        assert (up, down) == (4, 4), (up, down)
    finally:
        T.set_residency(True)


@pytest.mark.parametrize("zp,scale", [(120, 0.1), (-125, 1.0 / 7.0), (3, 0.3), (-128, 0.5), (127, 0.02), (10, 2.5)])
def test_dequantize_op_saturates_and_rounds_as_the_reference(zp, scale):
    """Round 6 (review item 4a) at the op boundary too: Register_DEQUANTIZE's int8 branch -- offset = TfLiteRound(1 / scale), a 0 bit ->
    min(127, zp + offset), a 1 bit -> max(-128, zp - offset) (tflite/kernels/quantization.cc:131-138) -- on zero points and scales that
    reach both saturating branches and reciprocals that round, against the oracle's unpack."""
    scale = float(np.float32(scale))
    for shape in ((1, 4, 4, 33), (2, 5, 3, 64), (1, 3, 3, 100)):
        words = synth.random_words(synth.rng(zp + shape[-1]), shape[:-1] + ((shape[-1] + 31) // 32,))
        d = T.SingleOpModel("LceDequantize")
        di = d.add_tensor(T.INT32, words.shape, words)
        do = d.add_tensor(T.INT8, shape, scale=scale, zero_point=zp)
        d.set_node([di], [do])
        assert d.prepare() == 0 and d.invoke() == 0, d.log
        assert np.array_equal(d.get(do), O.unpack(words, shape[-1], np.int8, scale=scale, zero_point=zp)), shape
