"""The comparison the reference's own op tests make (tflite/tests/bconv2d_test.cc:
606-742): run a plain float convolution on the same +-1 data, with one-padding
simulated by an explicit pad with +1 followed by a VALID convolution, apply the
fused activation, then post-multiply and post-add on the host."""
import numpy as np
import torch
import torch.nn.functional as F

import oracle_lib as O
import synth


def float_conv(spec: O.ConvSpec, inp_words, filt_words, post_mul, post_bias):
    cin_g = spec.channels_in // spec.groups
    x = synth.pm1_from_words(inp_words, spec.channels_in)            # [B,H,W,Cin]
    w = synth.pm1_from_words(filt_words, cin_g)                      # [Cout,KH,KW,Cin/G]
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).double()
    wt = torch.from_numpy(w).permute(0, 3, 1, 2).double()
    if spec.padding == O.PADDING_SAME:
        cs = spec.c_struct()
        pads = (cs.pad_w, cs.pad_w + cs.pad_w_offset, cs.pad_h, cs.pad_h + cs.pad_h_offset)
        xt = F.pad(xt, pads, value=1.0 if spec.pad_values == 1 else 0.0)
    y = F.conv2d(xt, wt, stride=(spec.stride_h, spec.stride_w),
                 dilation=(spec.dilation_h, spec.dilation_w), groups=spec.groups)
    y = y.permute(0, 2, 3, 1).numpy()                                # exact integers
    if spec.activation == O.ACT_RELU:
        y = np.maximum(y, 0)
    elif spec.activation == O.ACT_RELU6:
        y = np.clip(y, 0, 6)
    elif spec.activation == O.ACT_RELU_N1_TO_1:
        y = np.clip(y, -1, 1)
    return y, (y * post_mul.astype(np.float64) + post_bias.astype(np.float64))
