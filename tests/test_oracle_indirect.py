"""The oracle holds two restatements of the reference's portable CPU convolution: the direct loop
(core/bconv2d/reference.h:33-148) and the indirect BGEMM (core/indirect_bgemm/kernel.h:16-186 +
kernel_4x2_portable.h:22-158: packed 4-channel weight blocks, indirection table, 4x2 micro-kernel).  SURVEY.md
8(c) records that the reference's own two kernels are bit-identical on float / int8 outputs; so must these be --
on the reference's op-test grid shapes (tflite/tests/bconv2d_test.cc:790-856), odd pixel counts (the lone last
pixel of the 2-pixel blocks), channel counts that leave 1, 2 and 3 channels in the last block of four, groups."""
import numpy as np
import pytest

import oracle_lib as O
import synth


@pytest.mark.parametrize("inp,flt,groups", [
    ((1, 4, 4, 64), (3, 3, 4), 1), ((3, 8, 5, 64), (3, 3, 7), 1), ((5, 7, 7, 96), (2, 3, 5), 1),
    ((1, 7, 7, 192), (1, 1, 6), 2), ((1, 8, 5, 128), (3, 3, 64), 4), ((1, 7, 7, 512), (2, 3, 2), 2),
    ((2, 9, 3, 32), (3, 3, 1), 1), ((1, 5, 5, 256), (3, 3, 10), 2),
])
@pytest.mark.parametrize("stride,dilation", [((1, 1), (1, 1)), ((2, 3), (1, 1)), ((1, 1), (3, 2))])
@pytest.mark.parametrize("pad", ["VALID", "ONE"])
def test_indirect_formulation_equals_direct(inp, flt, groups, stride, dilation, pad):
    padding, pv = {"VALID": (O.PADDING_VALID, 0), "ONE": (O.PADDING_SAME, 1)}[pad]
    for act in (O.ACT_NONE, O.ACT_RELU):
        spec = O.ConvSpec(inp[0], inp[1], inp[2], inp[3], flt[0], flt[1], flt[2], groups, stride[0], stride[1],
                          dilation[0], dilation[1], padding, pv, act)
        if spec.out_h <= 0 or spec.out_w <= 0:
            continue
        x, w, mul, bias = synth.conv_inputs(spec, sum(inp) + sum(flt) + groups, negative_mul_fraction=0.2)
        a = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
        b = O.bconv2d_indirect(spec, O.DST_F32, x, w, mul, bias, threads=2)
        assert np.array_equal(a.view(np.int32), b.view(np.int32))
        scale, zp = synth.int8_quant_params(7)
        a = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=float(scale), out_zero_point=zp)
        b = O.bconv2d_indirect(spec, O.DST_I8, x, w, mul, bias, out_scale=float(scale), out_zero_point=zp)
        assert np.array_equal(a, b)


def test_indirect_formulation_refuses_zero_padding():
    spec = O.ConvSpec(1, 4, 4, 64, 3, 3, 4, padding=O.PADDING_SAME, pad_values=0)
    x, w, mul, bias = synth.conv_inputs(spec, 1)
    with pytest.raises(ValueError):
        O.bconv2d_indirect(spec, O.DST_F32, x, w, mul, bias)
