"""Pin the oracle's convolution the way the reference pins its own kernels: against
a float convolution on the same +-1 data, over the reference's test grid
(tflite/tests/bconv2d_test.cc:790-856; tolerances :374-428,767-768)."""
import itertools
import zlib

import numpy as np
import pytest

import oracle_lib as O
import synth
from float_conv_ref import float_conv

SMALL_INPUTS = [(1, 4, 4, c) for c in (4, 64, 96, 128, 192, 256)]
SMALL_FILTERS = [(1, 1, 1), (3, 3, 4), (3, 3, 64)]
BIG_INPUTS = [(1, 7, 7, 4), (3, 8, 5, 64), (5, 7, 7, 96), (1, 8, 5, 128), (1, 7, 7, 192),
              (1, 8, 5, 256), (1, 7, 7, 512)]
BIG_FILTERS = [(1, 1, 1), (3, 3, 1), (2, 3, 2), (1, 1, 3), (3, 3, 4), (2, 3, 5), (1, 1, 6),
               (3, 3, 7), (2, 3, 32)]
PADS = {"VALID": (O.PADDING_VALID, 0), "SAME": (O.PADDING_SAME, 0), "ONE": (O.PADDING_SAME, 1)}


def grid():
    for inp, flt, g in itertools.product(SMALL_INPUTS, SMALL_FILTERS, (1, 2)):
        for pad in ("VALID", "ONE"):
            yield inp, flt, g, (1, 1), (1, 1), pad, O.ACT_NONE
    for inp, flt, g, st, dil, pad, act in itertools.product(
            BIG_INPUTS, BIG_FILTERS, (1, 2, 4), ((1, 1), (2, 3)), ((1, 1), (3, 2)),
            ("VALID", "SAME", "ONE"), (O.ACT_NONE, O.ACT_RELU)):
        yield inp, flt, g, st, dil, pad, act
    # bconv2d_test.cc:813-828, the 16-bit-accumulator overflow shape
    for pad, act in itertools.product(("VALID", "ONE"), (O.ACT_NONE, O.ACT_RELU)):
        yield (1, 6, 6, 3072), (5, 5, 4), 1, (1, 1), (1, 1), pad, act


def legal(inp, flt, g, pad, sem):
    cin, cout = inp[3], flt[2]
    if g > 1 and (cin % g or cout % g or (cin // g) % 32):
        return False                      # bconv2d_test.cc:465-476
    if pad == "SAME" and sem == O.SEM_REFERENCE and cin % 2:
        return False                      # :499-504
    return True


CASES = [c for c in grid()]
# keep the CPU suite to minutes: every 3rd case of the big grid, all of the rest
CASES = CASES[:72] + CASES[72:-4:3] + CASES[-4:]


def _id(c):
    inp, flt, g, st, dil, pad, act = c
    return "I%s_K%s_G%d_S%dx%d_D%dx%d_%s_A%d" % ("x".join(map(str, inp)), "x".join(map(str, flt)),
                                                  g, st[0], st[1], dil[0], dil[1], pad, act)


@pytest.mark.parametrize("case", CASES, ids=_id)
def test_oracle_matches_float_conv(case):
    inp, flt, g, st, dil, pad, act = case
    for sem in (O.SEM_REFERENCE, O.SEM_OPTIMIZED):
        if not legal(inp, flt, g, pad, sem):
            continue
        padding, pad_values = PADS[pad]
        spec = O.ConvSpec(inp[0], inp[1], inp[2], inp[3], flt[0], flt[1], flt[2], g, st[0], st[1],
                          dil[0], dil[1], padding, pad_values, act, sem)
        if spec.out_h <= 0 or spec.out_w <= 0:
            continue
        x, w, mul, bias = synth.conv_inputs(spec, seed=zlib.crc32(_id(case).encode()) & 0xFFFF)
        conv, want = float_conv(spec, x, w, mul, bias)
        tol = 1e-2 if flt[0] * flt[1] * inp[3] > 4096 else 1e-3

        zero_pad = pad == "SAME"
        if not zero_pad:
            # exact: true dot product = a - 2 * accum (output_transform.h:62-91)
            acc = O.bconv2d_accum(spec, x, w)
            unclamped, _ = float_conv(O.ConvSpec(**{**spec.__dict__, "activation": O.ACT_NONE,
                                                   "_c": None}), x, w, mul, bias)
            assert np.array_equal(spec.backtransform_add - 2 * acc.astype(np.int64),
                                  unclamped.astype(np.int64))

        # float output (both semantics; optimized only legal with act NONE, bconv2d.cc:188-200)
        if not (zero_pad and sem == O.SEM_OPTIMIZED and act != O.ACT_NONE):
            got = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
            np.testing.assert_allclose(got, want, rtol=0, atol=tol)
            # far tighter than the reference asks: a few ulps of the folded bias (|bias'| <= ~1.5*(a+1))
            np.testing.assert_allclose(got, want, rtol=0, atol=4 * 2.0**-23 * 1.5 * (spec.backtransform_add + 2))

        if zero_pad and sem == O.SEM_OPTIMIZED:
            continue                       # int8 / bitpacked are rejected there (bconv2d.cc:188-200)

        # int8 output vs the UNROUNDED float (bconv2d_test.cc:408-428)
        scale, zp = synth.int8_quant_params(zlib.crc32(_id(case).encode()) & 0xFFF)
        got8 = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=float(scale), out_zero_point=zp)
        unrounded = np.clip(want / np.float64(scale) + zp, -128.0, 127.0)
        assert np.max(np.abs(got8.astype(np.float64) - unrounded)) <= (0.7 if tol == 1e-2 else 0.55)

        # bitpacked output: exact vs the sign of the float result (:374-393), thresholds
        # derived as the op test derives them (:327-368)
        thr = O.thresholds_optest(spec, mul, bias)
        gotb = O.bconv2d(spec, O.DST_BITPACKED, x, w, thresholds=thr)
        assert np.array_equal(gotb, O.bitpack(want.astype(np.float32)))
