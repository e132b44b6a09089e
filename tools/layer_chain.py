"""Device-resident chains of LceBconv2d layers (bench.py and the full-size GPU tests).

A chain = the binary convolutions of one BASELINE.json model stack (tools/synthetic_layers.py), each with its
own plan, seeded synthetic weights and HBM buffers.  Where a layer's output shape is the next layer's input
shape, the next layer consumes the previous one's LceQuantize (sign bits of a float output, or the int8
output quantized at its zero point); where the real network has a non-binary op in between (the section
transitions of QuickNet), the next layer starts from its own seeded input.  NumPy + torch + the C ABI only:
the oracle is not involved (tests compare against it separately).
"""
from __future__ import annotations

import numpy as np

import synthetic_layers as SL


class LayerChain:
    def __init__(self, amd, torch, layers, dev, dst="f32", seed=0, engine="auto"):
        self.amd, self.torch, self.layers, self.dev, self.dst = amd, torch, layers, dev, dst
        self.plans, self.x, self.y, self.bits, self.quant, self.weights = [], [], [], [], [], []
        adst = {"f32": amd.F32, "i8": amd.I8}[dst]
        for k, L in enumerate(layers):
            w, mul, bias, thr = SL.weights(L, seed + k)
            scale, zp = SL.int8_quant(seed + k) if dst == "i8" else (1.0, 0)
            plan = amd.Bconv2dPlan(L.params(amd, adst, scale, zp))
            plan.set_weights(w, mul, bias)
            plan.set_option("engine", engine)
            self.plans.append(plan)
            self.weights.append((w, mul, bias))
            self.quant.append((scale, zp))
            self.x.append(torch.from_numpy(SL.activations(L, seed + k)).to(dev))
            self.y.append(torch.empty(plan.output_shape, dtype=torch.float32 if dst == "f32" else torch.int8, device=dev))
            b, oh, ow, n = plan.output_shape
            self.bits.append(torch.empty((b, oh, ow, (n + 31) // 32), dtype=torch.int32, device=dev))
        # layer k + 1 is fed by layer k when the shapes line up
        self.fed = [False] + [tuple(self.bits[k].shape) == tuple(self.x[k + 1].shape) for k in range(len(layers) - 1)]

    @property
    def binary_macs(self):
        return sum(L.binary_macs for L in self.layers)

    def algorithmic_bytes(self):
        return sum(L.algorithmic_bytes(SL.DST_F32 if self.dst == "f32" else SL.DST_I8) for L in self.layers)

    def run_convs(self):
        """Every convolution on its own seeded input: the sum of the layers, no data dependence."""
        for p, x, y in zip(self.plans, self.x, self.y):
            p.run(x, y)

    def run_chain(self, fused=True):
        """The chain: output k is quantized into the input of layer k + 1 -- by the convolution's own epilogue when
        `fused` (float: sign bits; int8: q < output zero point), else by a separate LceQuantize pass."""
        amd = self.amd
        for k, p in enumerate(self.plans):
            x = self.bits[k - 1] if self.fed[k] else self.x[k]
            if self.dst == "f32":
                if fused:
                    p.run_dual(x, self.y[k], self.bits[k])
                else:
                    p.run(x, self.y[k])
                    amd.bitpack(self.y[k], out=self.bits[k])
            elif fused:
                p.run_dual(x, self.y[k], self.bits[k])       # int8 output + (q < zero point) bits from one epilogue
            else:
                p.run(x, self.y[k])
                amd.bitpack(self.y[k], self.quant[k][1], out=self.bits[k])

    def kernel_names(self, fused=False):
        return [p.kernel_name(dual=fused) for p in self.plans]
