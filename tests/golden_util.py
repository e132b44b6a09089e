import os

import numpy as np

import oracle_lib as O

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_FIELDS = ("batch", "in_h", "in_w", "channels_in", "filter_h", "filter_w", "channels_out", "groups",
           "stride_h", "stride_w", "dilation_h", "dilation_w", "padding", "pad_values", "activation",
           "semantics")


def conv_cases():
    z = np.load(os.path.join(_DIR, "bconv2d_golden.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    for n in names:
        spec = O.ConvSpec(**{f: int(v) for f, v in zip(_FIELDS, z[n + "/spec"])})
        d = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")}
        yield n, spec, d


def bitpack_cases():
    z = np.load(os.path.join(_DIR, "bitpack_golden.npz"))
    yield "f32_10x33", z["f32_10x33/in"], 0, z["f32_10x33/out"]
    yield "f32_4x6x6x64", z["f32_4x6x6x64/in"], 0, z["f32_4x6x6x64/out"]
    for zp in (-1000, -1, 0, 23, 127, 128):
        yield f"i8_15x63_zp{zp}", z["i8_15x63/in"], zp, z[f"i8_15x63_zp{zp}/out"]
    yield "bool_3x68", z["bool_3x68/in"], 0, z["bool_3x68/out"]
