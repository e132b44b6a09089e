"""Randomized binary sections through the whole stack above the C ABI (round 6): a `.tflite` of LCE ops is WRITTEN (tests/tflite_writer.py),
read back by the C++ reader, planned from the file, and run by `lce_tflite_model_run_section` -- true batches, the float / int8
`LceBconv2d` + `LceQuantize` pairs fused into one pass (`lce_hip_bconv2d_run_dual`), `LceBMaxPool2d`, `LceDequantize` -- against the
oracle run op by op (tflite/kernels/bconv2d.cc:550-564, quantization.cc:76-147, bmaxpool.cc:20-98).  The fixed models of
tests/test_gpu_model_runner.py have 64 / 96 / 40 / 33 channels; the draws here reach ragged channel counts, every output type as an
intermediate AND as a model output, strides, both paddings, activations and int8 zero points at the ends of the range -- the
combinations in which the round's randomized layer test found the one real bug of the round."""
import importlib
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, seed, settings
from hypothesis import strategies as st

import flexbuf
import oracle_lib as O
import synth
from test_model_reader_host import bconv_options
from tflite_writer import ModelBuilder

amd = importlib.import_module("compute-engine_amd")
mr = importlib.import_module("compute-engine_amd.model_runner")

pytestmark = pytest.mark.gpu


@st.composite
def _chain(draw):
    """A chain of 1-4 binary convolutions on a small map; per layer: output type, whether the tensor is also a model output, an optional
    2x2 binary max-pool behind a bitpacked output."""
    c = draw(st.sampled_from([32, 64, 70, 96, 160]))
    h, w = draw(st.integers(7, 14)), draw(st.integers(7, 14))
    layers = []
    ch, hh, ww = c, h, w
    for i in range(draw(st.integers(1, 4))):
        k = draw(st.sampled_from([1, 2, 3]))
        stride = draw(st.sampled_from([1, 1, 2]))
        pad = draw(st.sampled_from([(O.PADDING_VALID, 0), (O.PADDING_SAME, 1)]))
        if pad[0] == O.PADDING_VALID and (hh < k or ww < k):
            pad = (O.PADDING_SAME, 1)
        cout = draw(st.sampled_from([16, 32, 33, 40, 64, 80, 96, 128]))
        act = draw(st.sampled_from([O.ACT_NONE, O.ACT_RELU, O.ACT_RELU6]))
        dst = draw(st.sampled_from([O.DST_F32, O.DST_I8, O.DST_BITPACKED]))
        spec = O.ConvSpec(1, hh, ww, ch, k, k, cout, 1, stride, stride, 1, 1, pad[0], pad[1], act, O.SEM_OPTIMIZED)
        if spec.out_h < 1 or spec.out_w < 1:
            break
        pool = dst == O.DST_BITPACKED and spec.out_h >= 4 and spec.out_w >= 4 and draw(st.booleans())
        is_out = draw(st.booleans())
        layers.append(dict(spec=spec, dst=dst, pool=pool, is_out=is_out, seed=draw(st.integers(0, 9999))))
        ch, hh, ww = cout, spec.out_h // (2 if pool else 1), spec.out_w // (2 if pool else 1)
    return (c, h, w), layers, draw(st.sampled_from([1, 3, 8])), draw(st.integers(0, 9999))


def _build(shape, layers):
    c, h, w = shape
    b = ModelBuilder()
    t_in = b.tensor([1, h, w, c], np.float32, "input")
    t_bits = b.tensor([1, h, w, (c + 31) // 32], np.int32, "q0")
    b.custom_op("LceQuantize", [t_in], [t_bits], b"")
    outs, consts = [], []
    for i, L in enumerate(layers):
        s, dst = L["spec"], L["dst"]
        _, wts, mul, bias = synth.conv_inputs(s, L["seed"], negative_mul_fraction=0.2)
        scale, zp = synth.int8_quant_params(L["seed"])
        if L["seed"] % 5 == 0:
            zp = (-128, 127, -127)[(L["seed"] // 5) % 3]
        thr = O.thresholds_converter(s, mul, bias)
        t_w = b.tensor(wts.shape, np.int32, "w%d" % i, wts)
        n, oh, ow = s.channels_out, s.out_h, s.out_w
        if dst == O.DST_BITPACKED:
            t_t = b.tensor([n], np.int32, "thr%d" % i, thr)
            t_y = b.tensor([1, oh, ow, (n + 31) // 32], np.int32, "y%d" % i)
            b.custom_op("LceBconv2d", [t_bits, t_w, -1, -1, t_t], [t_y], bconv_options(s))
            t_next = t_y
            if L["pool"]:
                t_next = b.tensor([1, oh // 2, ow // 2, (n + 31) // 32], np.int32, "p%d" % i)
                b.custom_op("LceBMaxPool2d", [t_y], [t_next], flexbuf.bmaxpool_options(2, 2, 2, 2, O.PADDING_VALID))
            if L["is_out"] or i == len(layers) - 1:      # bits leave the model as floats: LceDequantize
                oh2, ow2 = (oh // 2, ow // 2) if L["pool"] else (oh, ow)
                t_d = b.tensor([1, oh2, ow2, n], np.float32, "d%d" % i)
                b.custom_op("LceDequantize", [t_next], [t_d], b"")
                outs.append(t_d)
        else:
            t_m = b.tensor([n], np.float32, "m%d" % i, mul)
            t_b = b.tensor([n], np.float32, "b%d" % i, bias)
            if dst == O.DST_F32:
                t_y = b.tensor([1, oh, ow, n], np.float32, "y%d" % i)
            else:
                t_y = b.tensor([1, oh, ow, n], np.int8, "y%d" % i, scale=float(scale), zero_point=zp)
            b.custom_op("LceBconv2d", [t_bits, t_w, t_m, t_b, -1], [t_y], bconv_options(s))
            if L["is_out"] or i == len(layers) - 1:
                outs.append(t_y)
            t_next = None
            if i < len(layers) - 1:
                t_next = b.tensor([1, oh, ow, (n + 31) // 32], np.int32, "q%d" % (i + 1))
                b.custom_op("LceQuantize", [t_y], [t_next], b"")
        consts.append(dict(w=wts, mul=mul, bias=bias, thr=thr, scale=float(scale), zp=zp))
        t_bits = t_next
    b.inputs, b.outputs = [t_in], outs
    return b.finish(), consts


def _oracle(x, layers, consts):
    n = x.shape[0]
    bits = O.bitpack(x)
    outs = []
    for i, (L, c) in enumerate(zip(layers, consts)):
        s, dst = L["spec"].with_batch(n), L["dst"]
        last = i == len(layers) - 1
        if dst == O.DST_BITPACKED:
            y = O.bconv2d(s, dst, bits, c["w"], thresholds=c["thr"])
            if L["pool"]:
                y = O.bmaxpool(y, 2, 2, 2, 2, O.PADDING_VALID)
            if L["is_out"] or last:
                outs.append(O.unpack(y, s.channels_out, np.float32))
            bits = y
        else:
            y = O.bconv2d(s, dst, bits, c["w"], c["mul"], c["bias"], out_scale=c["scale"], out_zero_point=c["zp"])
            if L["is_out"] or last:
                outs.append(y)
            bits = O.bitpack(y, c["zp"] if dst == O.DST_I8 else 0)
    return outs


_fused = [0]
_models = [0]


# (a longer hunt on the GPU box: LCE_FUZZ_EXAMPLES=5000 LCE_FUZZ_SEED=1 python -m pytest <this file> -- other draws than the suite's)
_FUZZ_N = int(os.environ.get("LCE_FUZZ_EXAMPLES", "80"))
_FUZZ_SEED = os.environ.get("LCE_FUZZ_SEED")


@(seed(int(_FUZZ_SEED)) if _FUZZ_SEED else (lambda f: f))
@settings(max_examples=_FUZZ_N, deadline=None, derandomize=_FUZZ_SEED is None, database=None,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large, HealthCheck.filter_too_much])
@given(_chain())
def test_random_binary_sections_equal_the_oracle_op_by_op(case):
    shape, layers, batch, seed = case
    if not layers:
        return
    data, consts = _build(shape, layers)
    c, h, w = shape
    n = batch + 2                                   # a ragged last batch
    x = synth.rng(seed).uniform(-1.5, 1.5, (n, h, w, c)).astype(np.float32)
    x[0, 0, 0, :4] = (-0.0, 0.0, np.nan, -np.inf)   # the sign test's special values (bitpack.h:72-110)
    want = _oracle(x, layers, consts)
    it = mr.Interpreter(data, batch_size=batch)
    got = it.predict(x)
    got = list(got) if isinstance(got, (list, tuple)) else [got]
    assert len(got) == len(want)
    for g, w_ in zip(got, want):
        assert g.shape == w_.shape and g.dtype == w_.dtype
        assert np.array_equal(g.view(np.uint8), w_.view(np.uint8)), [(L["spec"], L["dst"], L["pool"], L["is_out"]) for L in layers]
    _models[0] += 1
    _fused[0] += it.model.run_stats()[1]


def test_random_binary_sections_did_fuse_quantizes():
    """(same process, after the draws) the section runner fused LceBconv2d + LceQuantize pairs in many of the models."""
    if _models[0] == 0:
        pytest.skip("the randomized test did not run in this process")
    assert _models[0] >= min(60, _FUZZ_N - 20) and _fused[0] >= min(30, _FUZZ_N // 3), (_models, _fused)
