"""Pin the CPU oracle against every known-answer vector the reference's own tests
hold for this path (tests/golden/reference_kats.json)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def test_quantize_const_fold():
    for case in KATS["quantize_const_fold"]["cases"]:
        x = np.full(case["shape"], case["input_fill"], np.float32)
        got = O.bitpack(x)
        assert got.shape == tuple(case["shape"][:-1]) + (1,)
        assert got.ravel().tolist() == case["expected_words"]


def test_dequantize_const_fold():
    for case in KATS["dequantize_const_fold"]["cases"]:
        w = np.full(case["shape"], case["word"], np.int32)
        got = O.unpack(w, case["channels"], np.float32)
        assert np.all(got == np.float32(case["expected_fill"]))


def test_bitpack_weights():
    k = KATS["bitpack_weights"]
    got = O.bitpack(np.full(k["filter_shape_ohwi"], k["filter_fill"], np.float32))
    assert got.shape == tuple(k["expected_shape"])
    assert np.all(got == k["expected_fill"])


@pytest.mark.parametrize("blocks", KATS["one_hot_bit_order"]["num_4x32_blocks"])
@pytest.mark.parametrize("dtype", [np.float32, np.int8])
def test_one_hot_bit_order(blocks, dtype):
    k = KATS["one_hot_bit_order"]
    zp = k["zero_point_float"] if dtype == np.float32 else k["zero_point_int8"]
    n = 32 * 4 * blocks
    x = np.full((n, n), zp + 5, dtype)
    x[np.arange(n), np.arange(n)] = zp - 5          # row i is the one-hot vector e_i
    words = O.bitpack(x, zero_point=zp).view(np.uint32)
    bits = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(n, -1)
    assert np.array_equal(bits, np.eye(n, dtype=bits.dtype))


def test_converter_thresholds():
    k = KATS["converter_thresholds"]
    o, h, w, i = k["filter_shape_ohwi"]
    spec = O.ConvSpec(1, 4, 4, i, h, w, o, activation=k["activation"])
    got = O.thresholds_converter(spec, k["post_activation_multiplier"], k["post_activation_bias"])
    assert got.tolist() == k["expected_thresholds"]
