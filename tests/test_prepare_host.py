"""Converter-side parameter preparation (include/lce_hip.h, lce_hip_prepare_*; SURVEY.md 8(f)
row n2) against the reference converter's own known-answer vectors and against the oracle's
restatement of the same rules.  Host-only: runs without a GPU."""
import importlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import synth

amd = importlib.import_module("compute-engine_amd")
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def test_kat_thresholds_and_sign_flip_optimize_mlir():
    """mlir/tests/optimize.mlir:217-237: thresholds [0,3,2,2,INT32_MIN,2,1,2] and the first four
    filters (negative multipliers) flipped."""
    k = KATS["converter_thresholds"]
    cout, kh, kw, cin = k["filter_shape_ohwi"]
    f = np.ones((cout, kh, kw, cin), np.float32)
    flipped, thr = amd.prepare_bitpacked_output(f, k["post_activation_multiplier"], k["post_activation_bias"],
                                                k["activation"], amd.PADDING_SAME, 1)
    assert thr.tolist() == k["expected_thresholds"]
    assert np.all(flipped[:4] == -1.0) and np.all(flipped[4:] == 1.0)   # sign(0) counts as +


def test_kat_bitpack_weights_mlir():
    """mlir/tests/bitpack-weights.mlir:5,9: an all-ones 16x3x3x3 filter packs to zeros 16x3x3x1."""
    k = KATS["bitpack_weights"]
    words = amd.prepare_bitpack_filter(np.full(k["filter_shape_ohwi"], k["filter_fill"], np.float32))
    assert list(words.shape) == k["expected_shape"] and np.all(words == k["expected_fill"])
    # and the companion direction: all -1 -> every existing channel bit set, padding bits 0
    words = amd.prepare_bitpack_filter(np.full(k["filter_shape_ohwi"], -1.0, np.float32))
    assert np.all(words == 0b111)


@pytest.mark.parametrize("act", [O.ACT_NONE, O.ACT_RELU, O.ACT_RELU_N1_TO_1, O.ACT_RELU6])
def test_thresholds_match_oracle_rule(act):
    g = synth.rng(1000 + act)
    cout, kh, kw, cin = 257, 3, 3, 64
    mul = g.uniform(-1.5, 1.5, cout).astype(np.float32)
    bias = g.uniform(-40, 40, cout).astype(np.float32)
    mul[::17] = 0.0
    bias[::23] = 0.0
    f = np.ones((cout, kh, kw, cin), np.float32)
    _, thr = amd.prepare_bitpacked_output(f, mul, bias, act)
    spec = O.ConvSpec(1, 8, 8, cin, kh, kw, cout, activation=act)
    assert np.array_equal(thr, O.thresholds_converter(spec, mul, bias))


def test_prepare_binary_filter_transposes_and_scales():
    g = synth.rng(7)
    kh, kw, cin, cout = 3, 2, 40, 9
    sign = g.choice([-1.0, 1.0], (kh, kw, cin, cout)).astype(np.float32)
    scale = g.uniform(0.05, 3.0, cout).astype(np.float32)
    scale[3] *= -1                      # the converter takes |filter[0,0,0,o]|, whatever its sign
    hwio = sign * np.abs(scale)
    hwio[0, 0, 0, :] = np.abs(hwio[0, 0, 0, :]) * np.sign(scale)
    ohwi, mul, bias = amd.prepare_binary_filter(hwio)
    assert ohwi.shape == (cout, kh, kw, cin)
    assert np.array_equal(mul, np.abs(hwio[0, 0, 0, :])) and np.all(bias == 0)
    assert np.array_equal(ohwi, np.transpose(hwio / np.abs(hwio[0, 0, 0, :]), (3, 0, 1, 2)))
    assert set(np.unique(ohwi)) <= {-1.0, 1.0}
    # 0.5 % tolerance of IsBinaryFilter (prepare_tf.cc:66-92): 0.4 % off passes, 1 % off does not
    ok = hwio.copy()
    ok[1, 1, 5, 2] *= 1.004
    amd.prepare_binary_filter(ok)
    bad = hwio.copy()
    bad[1, 1, 5, 2] *= 1.01
    with pytest.raises(amd.LceHipError, match="not a binary filter"):
        amd.prepare_binary_filter(bad)
    zero = hwio.copy()
    zero[0, 0, 0, 4] = 0.0
    with pytest.raises(amd.LceHipError, match="zero scale"):
        amd.prepare_binary_filter(zero)


def test_fuse_post_ops_and_activation_rule():
    mul = np.array([1.0, 2.0, -0.5], np.float32)
    bias = np.array([0.0, 1.0, 4.0], np.float32)
    m, b = amd.prepare_fuse_post_op(amd.POST_ADD, [0.5, 0.25, -1.0], mul, bias)
    assert np.array_equal(m, mul) and np.array_equal(b, bias + np.float32([0.5, 0.25, -1.0]))
    m, b = amd.prepare_fuse_post_op(amd.POST_SUB, 2.0, mul, bias)
    assert np.array_equal(m, mul) and np.array_equal(b, bias - np.float32(2.0))
    m, b = amd.prepare_fuse_post_op(amd.POST_MUL, [3.0, -1.0, 0.1], mul, bias)
    assert np.array_equal(m, mul * np.float32([3.0, -1.0, 0.1])) and np.array_equal(b, bias * np.float32([3.0, -1.0, 0.1]))
    m, b = amd.prepare_fuse_post_op(amd.POST_DIV, 3.0, mul, bias)
    assert np.array_equal(m, mul / np.float32(3.0)) and np.array_equal(b, bias / np.float32(3.0))
    with pytest.raises(amd.LceHipError, match="scalar or have one entry per output channel"):
        amd.prepare_fuse_post_op(amd.POST_ADD, [1.0, 2.0], mul, bias)
    ones, zeros = np.ones(3, np.float32), np.zeros(3, np.float32)
    assert amd.prepare_can_fuse_activation(ones, zeros, amd.PADDING_VALID, 0)
    assert amd.prepare_can_fuse_activation(ones, zeros, amd.PADDING_SAME, 1)
    assert not amd.prepare_can_fuse_activation(ones, zeros, amd.PADDING_SAME, 0)
    assert not amd.prepare_can_fuse_activation(mul, zeros, amd.PADDING_VALID, 0)


def test_bitpacked_rewrite_only_for_the_two_converter_cases():
    f = np.ones((2, 1, 1, 32), np.float32)
    with pytest.raises(amd.LceHipError, match="only rewrites"):
        amd.prepare_bitpacked_output(f, [1, 1], [0, 0], padding=amd.PADDING_SAME, pad_values=0)


def test_bitpack_filter_matches_oracle_bitpack():
    g = synth.rng(3)
    f = g.choice([-1.0, 1.0], (5, 3, 3, 70)).astype(np.float32)
    f[0, 0, 0, 0] = -0.0                # -0.0 is not < 0: packs as +1 (bitpack.h:78)
    words = amd.prepare_bitpack_filter(f)
    want = O.bitpack(f.reshape(-1, 70)).reshape(words.shape)
    assert np.array_equal(words, want)
