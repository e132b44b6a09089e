"""The committed golden vectors must be reproduced by the oracle (no silent drift) and
by the host-simulated kernel bodies."""
import numpy as np
import pytest

import golden_util as G
import hostsim_lib as H
import oracle_lib as O


@pytest.mark.parametrize("name,spec,d", list(G.conv_cases()), ids=lambda v: v if isinstance(v, str) else "")
def test_conv_golden(name, spec, d):
    scale, zp = float(d["int8_scale_zp"][0]), int(d["int8_scale_zp"][1])
    x, w, mul, bias = d["input"], d["filter"], d["post_mul"], d["post_bias"]
    for impl in (lambda *a, **k: O.bconv2d(*a, **k), lambda *a, **k: H.bconv2d(*a, **k)[0]):
        got = impl(spec, O.DST_F32, x, w, mul, bias)
        assert np.array_equal(got.view(np.int32), d["out_f32"].view(np.int32))
        if "out_i8" in d:
            assert np.array_equal(impl(spec, O.DST_I8, x, w, mul, bias, out_scale=scale, out_zero_point=zp), d["out_i8"])
            assert np.array_equal(impl(spec, O.DST_BITPACKED, x, w, thresholds=d["thresholds"]), d["out_bitpacked"])


@pytest.mark.parametrize("name,x,zp,want", list(G.bitpack_cases()), ids=lambda v: v if isinstance(v, str) else "")
def test_bitpack_golden(name, x, zp, want):
    assert np.array_equal(O.bitpack(x, zp), want)
    assert np.array_equal(H.bitpack(x, zp), want)
