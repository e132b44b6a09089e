"""Independent pins of the conv oracle that do not go through the C oracle itself.

1. LITERAL int8 rounding: the portable reference rounds with std::round = half AWAY from zero and then saturates
   (core/bconv2d/output_transform.h:31-44).  Layers are constructed whose transformed value is EXACTLY +-0.5, +-1.5,
   +-2.5, 126.5, 127.5, -127.5, -128.5, ... and the expected int8 is written down by hand -- for the oracle and for the
   kernel bodies (hostsim: both engines).
2. A third, bit-level restatement of the accumulation in NumPy only -- unpack every word to +-1 (padding: +1 or 0),
   integer dot product over the window, accum = (K - dot) / 2 (core/types.h:45-47, output_transform.h:62-91) -- followed
   by the output transforms written with NumPy float32 arithmetic.  It recomputes every output stored in
   tests/golden/bconv2d_golden.npz, so the golden files (written by the C oracle) are checked by something that is not
   the code that wrote them.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as G   # noqa: E402
import hostsim_lib as H   # noqa: E402
import oracle_lib as O    # noqa: E402

# value before rounding -> int8 the reference produces: round half away from zero, then saturate to [-128, 127]
TIES = [(0.5, 1), (-0.5, -1), (1.5, 2), (-1.5, -2), (2.5, 3), (-2.5, -3), (0.0, 0), (0.25, 0), (-0.25, 0), (0.75, 1), (-0.75, -1),
        (126.5, 127), (127.5, 127), (127.0, 127), (-127.5, -128), (-128.5, -128), (-128.0, -128), (300.25, 127), (-300.75, -128),
        (3.5, 4), (4.5, 5), (-3.5, -4), (-4.5, -5), (63.5, 64), (-63.5, -64)]


def _tie_layer(zp):
    """A 1x1 layer over 32 channels whose output channel c has popcount c set filter bits: with an all-(+1) input word
    (0) the popcount accumulator of channel c is exactly c, and with post_mul = 1, scale = 2 the transformed value is
    -c + bias'_c; post_bias is chosen (exact in float) so that it equals TIES[c][0]."""
    n = len(TIES)
    spec = O.ConvSpec(1, 1, 1, 32, 1, 1, n)
    x = np.zeros(spec.input_shape(), np.int32)
    w = np.array([[(1 << c) - 1 if c < 32 else -1] for c in range(n)], np.uint32).view(np.int32).reshape(spec.filter_shape())
    scale, a = 2.0, 32.0
    mul = np.ones(n, np.float32)
    # bias' = (post_bias + a * post_mul) / scale + zp  and  y = -accum + bias'  ->  post_bias = 2 * (target + accum - zp) - a
    target = np.array([t for t, _ in TIES], np.float64)
    bias = (2.0 * (target + np.arange(n) - zp) - a).astype(np.float32)
    assert np.array_equal(bias.astype(np.float64), 2.0 * (target + np.arange(n) - zp) - a)     # exactly representable
    want = np.array([q for _, q in TIES], np.int8).reshape(1, 1, 1, n)
    return spec, x, w, mul, bias, scale, want


@pytest.mark.parametrize("zp", [0, 3, -7])
def test_int8_ties_round_half_away_then_saturate_literal_values(zp):
    spec, x, w, mul, bias, scale, want = _tie_layer(zp)
    got = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=scale, out_zero_point=zp)
    assert np.array_equal(got, want), (got.ravel().tolist(), want.ravel().tolist())
    got = O.bconv2d_indirect(spec, O.DST_I8, x, w, mul, bias, out_scale=scale, out_zero_point=zp)
    assert np.array_equal(got, want)
    for engine in ("valu", "mfma"):       # the kernel bodies, executed on the CPU
        got, name = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=scale, out_zero_point=zp, engine=engine)
        assert np.array_equal(got, want), (name, got.ravel().tolist())


# ----------------------------------------------------------------------------------------------------------------------
def _pm1(words, channels):
    """bit i of word k = channel 32k + i; 0 -> +1, 1 -> -1 (core/bitpacking/bitpack.h:72-110, 310-346)."""
    u = np.ascontiguousarray(words).view(np.uint32)
    bits = (u[..., :, None] >> np.arange(32, dtype=np.uint32)) & 1
    return (1 - 2 * bits.reshape(words.shape[:-1] + (-1,))[..., :channels].astype(np.int64))


def _same_pad(in_, k, stride, dil):
    """TFLite ComputeOutSize / ComputePaddingHeightWidth for SAME (tensorflow/lite/kernels/padding.h)."""
    out = (in_ + stride - 1) // stride
    total = max(0, (out - 1) * stride + (k - 1) * dil + 1 - in_)
    return out, total // 2


def numpy_accum(spec, x_words, f_words):
    """Popcount accumulators [B, OH, OW, Cout] from the +-1 integer dot product, nothing shared with the C oracle."""
    cin_g, cout_g = spec.channels_in // spec.groups, spec.channels_out // spec.groups
    x = _pm1(x_words, spec.channels_in)                       # [B, H, W, Cin]
    f = _pm1(f_words, cin_g)                                  # [Cout, KH, KW, Cin/G]
    eh, ew = (spec.filter_h - 1) * spec.dilation_h + 1, (spec.filter_w - 1) * spec.dilation_w + 1
    if spec.padding == O.PADDING_SAME:
        oh, ph = _same_pad(spec.in_h, spec.filter_h, spec.stride_h, spec.dilation_h)
        ow, pw = _same_pad(spec.in_w, spec.filter_w, spec.stride_w, spec.dilation_w)
    else:
        oh, ph = (spec.in_h + spec.stride_h - eh) // spec.stride_h, 0
        ow, pw = (spec.in_w + spec.stride_w - ew) // spec.stride_w, 0
    hp, wp = (oh - 1) * spec.stride_h + eh, (ow - 1) * spec.stride_w + ew
    fill = 1 if spec.pad_values == 1 else 0                   # one-padding: +1; zero-padding: the tap contributes 0
    xp = np.full((spec.batch, max(hp, ph + spec.in_h), max(wp, pw + spec.in_w), spec.channels_in), fill, np.int64)
    xp[:, ph:ph + spec.in_h, pw:pw + spec.in_w] = x
    dot = np.zeros((spec.batch, oh, ow, spec.channels_out), np.int64)
    for fy in range(spec.filter_h):
        for fx in range(spec.filter_w):
            ys, xs = fy * spec.dilation_h, fx * spec.dilation_w
            win = xp[:, ys:ys + (oh - 1) * spec.stride_h + 1:spec.stride_h, xs:xs + (ow - 1) * spec.stride_w + 1:spec.stride_w]
            for g in range(spec.groups):
                dot[..., g * cout_g:(g + 1) * cout_g] += np.einsum("bhwc,oc->bhwo", win[..., g * cin_g:(g + 1) * cin_g],
                                                                   f[g * cout_g:(g + 1) * cout_g, fy, fx])
    k = spec.filter_h * spec.filter_w * cin_g
    assert ((k - dot) % 2 == 0).all()
    return (k - dot) // 2


def _clamps(spec):
    """bconv2d.cc:380-388: the activation range in the back-transformed (+-1 sum) domain, expressed on 2 * accum."""
    a = spec.filter_h * spec.filter_w * (spec.channels_in // spec.groups)
    lo, hi = {O.ACT_NONE: (-a, a), O.ACT_RELU: (0, a), O.ACT_RELU6: (0, min(6, a)), O.ACT_RELU_N1_TO_1: (max(-1, -a), min(1, a))}[spec.activation]
    return -hi + a, -lo + a, a


def numpy_transform(spec, accum, post_mul, post_bias, scale=1.0, zp=0.0):
    """output_transform.h:93-107 with the OneTimeSetup folding of bconv2d.cc:364-378 (double -> float), two roundings."""
    cmin, cmax, a = _clamps(spec)
    mul = (-1.0 * post_mul.astype(np.float64) / scale).astype(np.float32)
    bias = ((post_bias.astype(np.float64) + a * post_mul.astype(np.float64)) / scale + zp).astype(np.float32)
    xf = np.clip(2 * accum, cmin, cmax).astype(np.float32)
    prod = (xf * mul).astype(np.float32)                      # first rounding
    return (prod + bias).astype(np.float32)                   # second rounding


def numpy_round_sat_i8(y):
    """std::round (half away from zero) in float, then saturate (output_transform.h:31-44)."""
    y = y.astype(np.float64)
    r = np.where(y >= 0, np.floor(y + 0.5), np.ceil(y - 0.5))
    return np.clip(r, -128, 127).astype(np.int8)


@pytest.mark.parametrize("name,spec,d", list(G.conv_cases()), ids=lambda v: v if isinstance(v, str) else "")
def test_golden_conv_outputs_recomputed_without_the_c_oracle(name, spec, d):
    if spec.padding == O.PADDING_SAME and spec.pad_values == 0 and spec.semantics == O.SEM_OPTIMIZED:
        pytest.skip("the optimized kernels' float correction of SAME-zero padding is pinned in test_oracle_vs_float_conv.py")
    accum = numpy_accum(spec, d["input"], d["filter"])
    y = numpy_transform(spec, accum, d["post_mul"], d["post_bias"])
    assert np.array_equal(y.view(np.int32), d["out_f32"].view(np.int32)), name
    scale, zp = float(d["int8_scale_zp"][0]), int(d["int8_scale_zp"][1])
    q = numpy_round_sat_i8(numpy_transform(spec, accum, d["post_mul"], d["post_bias"], scale, zp))
    assert np.array_equal(q, d["out_i8"]), name
    bits = accum > d["thresholds"].astype(np.int64)            # output_transform.h:160-168
    pad = (-spec.channels_out) % 32
    bits = np.concatenate([bits, np.zeros(bits.shape[:-1] + (pad,), bool)], axis=-1)
    words = (bits.reshape(bits.shape[:-1] + (-1, 32)).astype(np.uint32) << np.arange(32, dtype=np.uint32)).sum(-1).astype(np.uint32)
    assert np.array_equal(words.view(np.int32), d["out_bitpacked"]), name
