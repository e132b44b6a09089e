"""Seeded synthetic inputs for the LceBconv2d path.

Mirrors the reference's own test data: +-1 activations/weights as Bernoulli(0.5)
bits, post-multiplier / post-bias ~ U(0.01, 1.5)
(tflite/tests/bconv2d_test.cc:570-581), int8 output scale 1/n with n in 1..20 and
zero point in [-20, 20] (tflite/tests/utils.h:58-62).
"""
from __future__ import annotations

import numpy as np

import oracle_lib as O

SEED_BASE = 0x1CE0000


def rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(SEED_BASE + seed))


def random_words(g: np.random.Generator, shape, valid_bits_last: int | None = None) -> np.ndarray:
    """Random bitpacked words; if valid_bits_last is given, the padding bits of the
    last word along the final axis are cleared (as LceQuantize guarantees)."""
    w = g.integers(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
    if valid_bits_last is not None and valid_bits_last % 32:
        mask = np.uint32((1 << (valid_bits_last % 32)) - 1)
        w[..., -1] &= mask
    return w.view(np.int32)


def conv_inputs(spec: O.ConvSpec, seed: int, negative_mul_fraction: float = 0.0):
    g = rng(seed)
    cin_g = spec.channels_in // spec.groups
    inp = random_words(g, spec.input_shape(), spec.channels_in)
    filt = random_words(g, spec.filter_shape(), cin_g)
    post_mul = g.uniform(0.01, 1.5, spec.channels_out).astype(np.float32)
    post_bias = g.uniform(0.01, 1.5, spec.channels_out).astype(np.float32)
    if negative_mul_fraction > 0:
        neg = g.random(spec.channels_out) < negative_mul_fraction
        post_mul[neg] *= -1
    return inp, filt, post_mul, post_bias


def int8_quant_params(seed: int):
    g = rng(seed ^ 0x5A5A)
    n = int(g.integers(1, 21))
    return np.float32(1.0) / np.float32(n), int(g.integers(-20, 21))


def pm1_from_words(words: np.ndarray, channels: int) -> np.ndarray:
    """Bit 0 -> +1.0, bit 1 -> -1.0 (core/bitpacking/bitpack.h:310-346)."""
    u = words.view(np.uint32)
    bits = (u[..., :, None] >> np.arange(32, dtype=np.uint32)) & 1
    bits = bits.reshape(words.shape[:-1] + (-1,))[..., :channels]
    return (1.0 - 2.0 * bits).astype(np.float32)
