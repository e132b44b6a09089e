"""Synthetic LceBconv2d layers for bench.py and tools/ -- NumPy only, no oracle.

The oracle (oracle/, tests/oracle_lib.py) is test infrastructure: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may touch it.  Everything else that needs
a layer description and seeded inputs (SURVEY.md 8(d) "Synthetic inputs": PCG64, seed
0x1CE0000 + n, Bernoulli(0.5) bits, post_mul / post_bias ~ U(0.01, 1.5)) gets them here."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

PADDING_SAME, PADDING_VALID = 0, 1      # tflite schema Padding
ACT_NONE, ACT_RELU = 0, 1
DST_F32, DST_I8, DST_BITPACKED = "f32", "i8", "bitpacked"


def rng(n: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(0x1CE0000 + int(n)))


def random_words(g: np.random.Generator, shape, channels: int) -> np.ndarray:
    """int32 words of i.i.d. fair bits; bits past `channels` in the last word are 0."""
    w = g.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32)
    if channels % 32:
        w[..., -1] &= np.uint32((1 << (channels % 32)) - 1)
    return w.view(np.int32)


@dataclass(frozen=True)
class Layer:
    batch: int
    in_h: int
    in_w: int
    channels_in: int
    filter_h: int
    filter_w: int
    channels_out: int
    stride: int = 1
    padding: int = PADDING_VALID
    pad_values: int = 0
    activation: int = ACT_NONE

    @property
    def in_words(self):
        return (self.channels_in + 31) // 32

    @property
    def out_words(self):
        return (self.channels_out + 31) // 32

    def _out(self, n, k):   # TFLite ComputeOutSize
        return (n + self.stride - 1) // self.stride if self.padding == PADDING_SAME else (n - k + self.stride) // self.stride

    @property
    def out_h(self):
        return self._out(self.in_h, self.filter_h)

    @property
    def out_w(self):
        return self._out(self.in_w, self.filter_w)

    @property
    def binary_macs(self) -> int:
        """B*OH*OW*Cout*KH*KW*Cin (padded taps included, as the reference counts)."""
        return self.batch * self.out_h * self.out_w * self.channels_out * self.filter_h * self.filter_w * self.channels_in

    def input_shape(self):
        return (self.batch, self.in_h, self.in_w, self.in_words)

    def filter_shape(self):
        return (self.channels_out, self.filter_h, self.filter_w, self.in_words)

    def algorithmic_bytes(self, dst: str) -> int:
        """input words + weights + per-channel params + output, each counted once."""
        inp = self.batch * self.in_h * self.in_w * self.in_words * 4
        wts = self.channels_out * self.filter_h * self.filter_w * self.in_words * 4
        pix = self.batch * self.out_h * self.out_w
        if dst == DST_BITPACKED:
            return inp + wts + self.channels_out * 4 + pix * self.out_words * 4
        return inp + wts + self.channels_out * 8 + pix * self.channels_out * (4 if dst == DST_F32 else 1)

    def params(self, amd, dst_type, out_scale=1.0, out_zero_point=0):
        """The C-ABI descriptor (compute-engine_amd.ConvParams) of this layer."""
        return amd.ConvParams(self.batch, self.in_h, self.in_w, self.channels_in, self.filter_h, self.filter_w,
                              self.channels_out, 1, self.stride, self.stride, 1, 1, self.padding, self.pad_values,
                              self.activation, dst_type, amd.SEM_OPTIMIZED, float(out_scale), int(out_zero_point))


def weights(layer: Layer, seed: int):
    """(filter words OHWI, post_activation_multiplier, post_activation_bias, thresholds)."""
    g = rng(seed)
    filt = random_words(g, layer.filter_shape(), layer.channels_in)
    mul = g.uniform(0.01, 1.5, layer.channels_out).astype(np.float32)
    bias = g.uniform(0.01, 1.5, layer.channels_out).astype(np.float32)
    # thresholds spread around the middle of the accumulator range so both bit values occur
    a = layer.filter_h * layer.filter_w * layer.channels_in
    thr = (a // 2 + g.integers(-a // 16 - 1, a // 16 + 2, layer.channels_out)).astype(np.int32)
    return filt, mul, bias, thr


def activations(layer: Layer, seed: int) -> np.ndarray:
    return random_words(rng(seed + 7), layer.input_shape(), layer.channels_in)


# ---- the BASELINE.json model stacks, as lists of layers ----------------
QUICKNET_STAGES = ((56, 64), (28, 128), (14, 256), (7, 512))      # (H = W, channels) of the four binary sections


def quicknet_layers(batch: int, blocks=(4, 4, 4, 4)) -> list:
    """QuickNet's binary convolutions (3x3, stride 1, SAME with pad_values = 1, float output feeding a
    residual add): `blocks` layers per section -- (4, 4, 4, 4) = QuickNet, (6, 8, 12, 6) = QuickNetLarge."""
    out = []
    for n, (hw, c) in zip(blocks, QUICKNET_STAGES):
        out += [Layer(batch, hw, hw, c, 3, 3, c, padding=PADDING_SAME, pad_values=1) for _ in range(n)]
    return out


def birealnet_layers(batch: int) -> list:
    """a Bi-RealNet-style chain with int8 outputs and a fused RELU -- per section
    1x1 stride 1 -> 3x3 stride 1 -> 3x3 stride 2 (which also doubles the channels for the next section)."""
    out = []
    for k, (hw, c) in enumerate(QUICKNET_STAGES):
        nxt = QUICKNET_STAGES[k + 1][1] if k + 1 < len(QUICKNET_STAGES) else c
        out.append(Layer(batch, hw, hw, c, 1, 1, c, activation=ACT_RELU))
        out.append(Layer(batch, hw, hw, c, 3, 3, c, padding=PADDING_SAME, pad_values=1, activation=ACT_RELU))
        out.append(Layer(batch, hw, hw, c, 3, 3, nxt, stride=2, padding=PADDING_SAME, pad_values=1, activation=ACT_RELU))
    return out


def int8_quant(seed: int):
    """(output scale, output zero point) of an int8-output layer: scale in [1/16, 1/4], zero point in [-20, 20]."""
    g = rng(seed + 31)
    return float(g.choice([1 / 16, 1 / 8, 3 / 16, 1 / 4])), int(g.integers(-20, 21))
