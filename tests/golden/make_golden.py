#!/usr/bin/env python
"""Regenerates tests/golden/bconv2d_golden.npz and bitpack_golden.npz.

The vectors are produced by the CPU oracle (oracle/lce_oracle.c), which is itself pinned
against the reference's known-answer tests (reference_kats.json) and against the
float-convolution property the reference's op tests use.  The reference cannot be built
or imported in this image (un-vendored TensorFlow Lite / Ruy), so there are no
reference-generated conv outputs to store; see DESIGN.md "Oracle pinning".

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import synth            # noqa: E402

CONV_CASES = {
    # name: (ConvSpec kwargs)
    "l0_like_same_one": dict(batch=2, in_h=6, in_w=7, channels_in=256, filter_h=3, filter_w=3,
                             channels_out=64, padding=O.PADDING_SAME, pad_values=1),
    "quicknet64_relu": dict(batch=1, in_h=8, in_w=8, channels_in=64, filter_h=3, filter_w=3,
                            channels_out=64, padding=O.PADDING_SAME, pad_values=1,
                            activation=O.ACT_RELU),
    "pointwise_odd_words": dict(batch=3, in_h=5, in_w=4, channels_in=96, filter_h=1, filter_w=1,
                                channels_out=40),
    "grouped_strided_dilated": dict(batch=2, in_h=9, in_w=8, channels_in=128, filter_h=2, filter_w=3,
                                    channels_out=32, groups=4, stride_h=2, stride_w=3, dilation_h=3,
                                    dilation_w=2, padding=O.PADDING_SAME, pad_values=1),
    "ragged_channels_valid": dict(batch=1, in_h=7, in_w=7, channels_in=20, filter_h=3, filter_w=3,
                                  channels_out=7),
    "same_zero_reference": dict(batch=1, in_h=6, in_w=6, channels_in=64, filter_h=3, filter_w=3,
                                channels_out=33, padding=O.PADDING_SAME, pad_values=0,
                                semantics=O.SEM_REFERENCE),
    "same_zero_optimized": dict(batch=2, in_h=7, in_w=5, channels_in=64, filter_h=3, filter_w=3,
                                channels_out=16, stride_h=2, stride_w=1, padding=O.PADDING_SAME,
                                pad_values=0, semantics=O.SEM_OPTIMIZED),
}


def main():
    out = {}
    for idx, (name, kw) in enumerate(CONV_CASES.items()):
        spec = O.ConvSpec(**kw)
        x, w, mul, bias = synth.conv_inputs(spec, 1000 + idx, negative_mul_fraction=0.0)
        scale, zp = synth.int8_quant_params(1000 + idx)
        out[name + "/spec"] = np.array([getattr(spec, f) for f in (
            "batch", "in_h", "in_w", "channels_in", "filter_h", "filter_w", "channels_out", "groups",
            "stride_h", "stride_w", "dilation_h", "dilation_w", "padding", "pad_values", "activation",
            "semantics")], np.int32)
        out[name + "/input"], out[name + "/filter"] = x, w
        out[name + "/post_mul"], out[name + "/post_bias"] = mul, bias
        out[name + "/int8_scale_zp"] = np.array([scale, zp], np.float32)
        out[name + "/out_f32"] = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
        opt_zero = (spec.padding == O.PADDING_SAME and spec.pad_values == 0
                    and spec.semantics == O.SEM_OPTIMIZED)
        if not opt_zero:
            thr = O.thresholds_converter(spec, mul, bias)
            out[name + "/thresholds"] = thr
            out[name + "/out_i8"] = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=float(scale),
                                              out_zero_point=int(zp))
            out[name + "/out_bitpacked"] = O.bconv2d(spec, O.DST_BITPACKED, x, w, thresholds=thr)
    np.savez_compressed(os.path.join(HERE, "bconv2d_golden.npz"), **out)

    g = synth.rng(4242)
    bp = {}
    f = g.uniform(-1.5, 1.5, (10, 33)).astype(np.float32)
    f[0, :4] = [-0.0, 0.0, np.nan, -np.inf]
    bp["f32_10x33/in"], bp["f32_10x33/out"] = f, O.bitpack(f)
    f2 = g.uniform(-1.5, 1.5, (4, 6, 6, 64)).astype(np.float32)
    bp["f32_4x6x6x64/in"], bp["f32_4x6x6x64/out"] = f2, O.bitpack(f2)
    i8 = g.integers(-128, 128, (15, 63)).astype(np.int8)
    for zp in (-1000, -1, 0, 23, 127, 128):
        bp[f"i8_15x63_zp{zp}/out"] = O.bitpack(i8, zp)
    bp["i8_15x63/in"] = i8
    b = g.integers(0, 2, (3, 68)).astype(np.bool_)
    bp["bool_3x68/in"], bp["bool_3x68/out"] = b, O.bitpack(b)
    np.savez_compressed(os.path.join(HERE, "bitpack_golden.npz"), **bp)
    for fn in ("bconv2d_golden.npz", "bitpack_golden.npz"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")


if __name__ == "__main__":
    main()
