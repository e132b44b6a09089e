"""CPU execution of the real kernel bodies + real planner against the oracle.

These tests do not replace the GPU parity tests (tests/test_gpu_*.py): they only prove
that the kernels' index arithmetic, padding, grouping, tiling and fused epilogues are
right before GPU minutes are spent."""
import itertools
import zlib

import ctypes as C

import numpy as np
import pytest

import hostsim_lib as H
import oracle_lib as O
import synth
from test_oracle_vs_float_conv import CASES, PADS, _id, legal

TILES = [(4, 16), (2, 32), (2, 16), (1, 32), (1, 16)]


def test_fastdiv_exhaustive_small_and_edges():
    rs = np.random.default_rng(0)
    for d in list(range(1, 70)) + [112, 224, 1000, 3136, 65535, 65536, 2**20 + 7, 2**31 - 1]:
        ns = np.concatenate([np.arange(0, 300), rs.integers(0, 2**31, 200), [2**31 - 1, 2**31 - 2, d - 1, d, d + 1]])
        ns = ns[(ns >= 0) & (ns < 2**31)]
        for n in ns:
            assert H.fastdiv(int(n), d) == int(n) // d, (n, d)


def _run_all_dst(spec, seed, kernel, tile, max_batch=0):
    x, w, mul, bias = synth.conv_inputs(spec, seed)
    zero_pad = spec.padding == O.PADDING_SAME and spec.pad_values == 0
    ran = []
    # float
    if not (zero_pad and spec.semantics == O.SEM_OPTIMIZED and spec.activation != O.ACT_NONE):
        want = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
        got, name = H.bconv2d(spec, O.DST_F32, x, w, mul, bias, kernel=kernel, tile=tile, max_batch=max_batch)
        assert np.array_equal(got.view(np.int32), want.view(np.int32)), name   # bit-exact floats
        ran.append(name)
    if zero_pad and spec.semantics == O.SEM_OPTIMIZED:
        return ran
    scale, zp = synth.int8_quant_params(seed)
    want = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=float(scale), out_zero_point=zp)
    got, name = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=float(scale), out_zero_point=zp,
                          kernel=kernel, tile=tile, max_batch=max_batch)
    assert np.array_equal(got, want), name
    ran.append(name)
    thr = O.thresholds_converter(spec, mul, bias)
    thr[::5] = np.iinfo(np.int32).max
    thr[1::7] = np.iinfo(np.int32).min
    want = O.bconv2d(spec, O.DST_BITPACKED, x, w, thresholds=thr)
    if tile[1] in (0, 32):
        got, name = H.bconv2d(spec, O.DST_BITPACKED, x, w, thresholds=thr, kernel=kernel, tile=tile,
                              max_batch=max_batch)
        assert np.array_equal(got, want), name
        ran.append(name)
    return ran


def _spec(case, sem):
    inp, flt, g, st, dil, pad, act = case
    padding, pad_values = PADS[pad]
    return O.ConvSpec(inp[0], inp[1], inp[2], inp[3], flt[0], flt[1], flt[2], g, st[0], st[1],
                      dil[0], dil[1], padding, pad_values, act, sem)


GRID = CASES[:72] + CASES[72::7]


@pytest.mark.parametrize("case", GRID, ids=_id)
def test_reference_grid_auto_and_general(case):
    """The reference's own shape grid (bconv2d_test.cc:790-856), both registrations'
    semantics, through the planner's automatic choice and through the general kernel."""
    for sem in (O.SEM_REFERENCE, O.SEM_OPTIMIZED):
        if not legal(case[0], case[1], case[2], case[5], sem):
            continue
        spec = _spec(case, sem)
        if spec.out_h <= 0 or spec.out_w <= 0:
            continue
        seed = zlib.crc32(_id(case).encode()) & 0xFFFF
        _run_all_dst(spec, seed, "auto", (0, 0))
        _run_all_dst(spec, seed, "general", (0, 0))


@pytest.mark.parametrize("tile", TILES, ids=lambda t: "%dx%d" % t)
@pytest.mark.parametrize("cin,cout,groups", [(64, 64, 1), (32, 40, 1), (96, 33, 1), (128, 64, 2),
                                             (256, 128, 4), (20, 7, 1), (160, 96, 1)])
@pytest.mark.parametrize("pad", ["VALID", "SAME", "ONE"])
def test_every_tile_shape(tile, cin, cout, groups, pad):
    """Every instantiated (TM, TN) x CH variant, ragged pixel counts (M not a multiple of
    64*TM), ragged channel tiles, odd word counts, strides and dilation."""
    if groups > 1 and (cout // groups) % tile[1]:
        pytest.skip("tile would straddle groups; planner falls back (covered elsewhere)")
    padding, pad_values = PADS[pad]
    for sem, st, dil, act in [(O.SEM_REFERENCE, (1, 1), (1, 1), O.ACT_NONE),
                              (O.SEM_OPTIMIZED, (2, 1), (1, 2), O.ACT_NONE),
                              (O.SEM_REFERENCE, (1, 2), (2, 1), O.ACT_RELU)]:
        if pad == "SAME" and sem == O.SEM_REFERENCE and cin % 2:
            continue
        spec = O.ConvSpec(3, 9, 11, cin, 3, 3, cout, groups, st[0], st[1], dil[0], dil[1], padding,
                          pad_values, act, sem)
        names = _run_all_dst(spec, seed=cin * 7 + cout, kernel="tiled", tile=tile)
        assert all("TM=%d,TN=%d" % tile in n for n in names), names


def test_batch_chunking_matches_single_launch():
    spec = O.ConvSpec(5, 6, 7, 64, 3, 3, 32, padding=O.PADDING_SAME, pad_values=1)
    _run_all_dst(spec, 11, "auto", (0, 0), max_batch=2)
    _run_all_dst(spec, 11, "tiled", (1, 32), max_batch=1)


@pytest.mark.parametrize("act", [O.ACT_NONE, O.ACT_RELU, O.ACT_RELU_N1_TO_1, O.ACT_RELU6])
def test_all_fused_activations(act):
    spec = O.ConvSpec(2, 5, 5, 64, 3, 3, 48, padding=O.PADDING_SAME, pad_values=1, activation=act)
    _run_all_dst(spec, 5 + act, "auto", (0, 0))


def test_int8_exact_ties_round_half_away():
    """y = n + 0.5 exactly: the portable reference rounds half AWAY from zero
    (output_transform.h:31-44 -> std::round), not half-to-even."""
    spec = O.ConvSpec(1, 4, 4, 32, 1, 1, 16)
    x, w, _, _ = synth.conv_inputs(spec, 3)
    mul = np.full(16, -0.25, np.float32)        # folded mul = +0.25 -> y = acc/2*... quarter steps
    bias = np.linspace(-8, 7, 16).astype(np.float32)
    want = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=1.0, out_zero_point=0)
    got, _ = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=1.0, out_zero_point=0)
    assert np.array_equal(got, want)
    f = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
    assert np.any(np.abs(f - np.trunc(f)) == 0.5), "test must contain exact ties"


@pytest.mark.parametrize("dtype", [np.float32, np.int8, np.bool_])
@pytest.mark.parametrize("rows,cols", [(1, 1), (2, 3), (3, 16), (8, 32), (10, 33), (15, 63), (64, 64),
                                       (7, 128), (3, 200), (33, 32), (5, 1024), (70, 96)])
def test_bitpack_kernels(dtype, rows, cols):
    g = synth.rng(rows * 1000 + cols)
    if dtype == np.float32:
        x = g.uniform(-1.5, 1.5, (rows, cols)).astype(np.float32)
        x[0, 0] = -0.0
        if cols > 2:
            x[0, 1] = np.nan
        zps = [0]
    elif dtype == np.int8:
        x = g.integers(-128, 128, (rows, cols)).astype(np.int8)
        zps = [-1000, -128, -1, 0, 23, 127, 128]
    else:
        x = g.integers(0, 2, (rows, cols)).astype(np.bool_)
        zps = [1]
    for zp in zps:
        want = O.bitpack(x, zp if dtype == np.int8 else 0)
        assert np.array_equal(H.bitpack(x, zp), want)
        assert np.array_equal(H.bitpack(x, zp, force_rows=True), want)


@pytest.mark.parametrize("cols", [1, 2, 31, 32, 33, 64, 68, 96, 256])
def test_quantize_dequantize_round_trip(cols):
    """tflite/tests/quantization_test.cc:75-130."""
    g = synth.rng(cols)
    shape = (1, 4, 4, cols)
    signs = np.where(g.random(shape) < 0.5, -1.0, 1.0).astype(np.float32)
    assert np.array_equal(H.unpack(H.bitpack(signs), cols, np.float32), signs)
    assert np.array_equal(H.unpack(O.bitpack(signs), cols, np.float32), O.unpack(O.bitpack(signs), cols, np.float32))
    n, zp = int(g.integers(1, 21)), int(g.integers(-20, 21))
    q = (zp + n * signs).astype(np.int8)
    back = H.unpack(H.bitpack(q, zp), cols, np.int8, scale=np.float32(1.0) / np.float32(n), zero_point=zp)
    assert np.array_equal(back, q)
    b = signs > 0
    assert np.array_equal(H.unpack(H.bitpack(b), cols, np.bool_), b)


@pytest.mark.parametrize("f,s,pad", [((2, 2), (2, 2), O.PADDING_SAME), ((3, 3), (2, 2), O.PADDING_SAME),
                                     ((3, 2), (1, 2), O.PADDING_VALID), ((2, 3), (3, 1), O.PADDING_SAME)])
def test_bmaxpool(f, s, pad):
    g = synth.rng(99)
    for words in (3, 8):      # one word per thread, and the 16-byte path (words % 4 == 0)
        x = synth.random_words(g, (2, 9, 7, words))
        assert np.array_equal(H.bmaxpool(x, f[0], f[1], s[0], s[1], pad), O.bmaxpool(x, f[0], f[1], s[0], s[1], pad))


def test_round_sat_i8_every_float():
    """The int8 epilogue's rounding (clamp, add copysign(pred(0.5)), truncate) equals the
    reference's saturate(std::round(y)) (output_transform.h:31-44) for every float."""
    bad, first = H.check_round_sat_i8()
    assert bad == 0, "first mismatch at float bits 0x%08x" % first


# ------------------------------------------------------------------------------------ matrix-core engine

MFMA_TILES = [(256, 256), (256, 128), (512, 64), (128, 256), (128, 128), (256, 64), (128, 64)]


def _run_all_dst_mfma(spec, seed, tile=(0, 0), max_batch=0, engine="mfma"):
    x, w, mul, bias = synth.conv_inputs(spec, seed, negative_mul_fraction=0.2)
    zero_pad = spec.padding == O.PADDING_SAME and spec.pad_values == 0
    names = []
    if not (zero_pad and spec.semantics == O.SEM_OPTIMIZED and spec.activation != O.ACT_NONE):
        want = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
        got, name = H.bconv2d(spec, O.DST_F32, x, w, mul, bias, tile=tile, max_batch=max_batch, engine=engine)
        assert np.array_equal(got.view(np.int32), want.view(np.int32)), name
        names.append(name)
    if zero_pad and spec.semantics == O.SEM_OPTIMIZED:
        return names
    scale, zp = synth.int8_quant_params(seed)
    want = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=float(scale), out_zero_point=zp)
    got, name = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=float(scale), out_zero_point=zp,
                          tile=tile, max_batch=max_batch, engine=engine)
    assert np.array_equal(got, want), name
    names.append(name)
    thr = O.thresholds_converter(spec, mul, bias)
    thr[::5] = np.iinfo(np.int32).max
    thr[1::7] = np.iinfo(np.int32).min
    thr[2::11] = -1
    want = O.bconv2d(spec, O.DST_BITPACKED, x, w, thresholds=thr)
    got, name = H.bconv2d(spec, O.DST_BITPACKED, x, w, thresholds=thr, tile=tile, max_batch=max_batch, engine=engine)
    assert np.array_equal(got, want), name
    names.append(name)
    return names


@pytest.mark.parametrize("tile", MFMA_TILES, ids=lambda t: "%dx%d" % t)
@pytest.mark.parametrize("cin,cout,pad", [(64, 64, "ONE"), (96, 33, "SAME"), (20, 7, "VALID"), (160, 130, "ONE")])
def test_mfma_engine_every_tile(tile, cin, cout, pad):
    """The FP4 matrix-core engine (expand_fp4 + bconv2d_mfma, MFMA emulated lane-exactly in
    the host simulation) against the oracle: ragged pixel/channel tiles, channel counts
    that are not multiples of 64, strides, dilation, both zero-padding semantics."""
    padding, pad_values = PADS[pad]
    combos = {"ONE": [(O.SEM_REFERENCE, (1, 1), (1, 1), O.ACT_NONE)],
              "SAME": [(O.SEM_OPTIMIZED, (2, 1), (1, 2), O.ACT_NONE), (O.SEM_REFERENCE, (1, 1), (1, 1), O.ACT_RELU)],
              "VALID": [(O.SEM_REFERENCE, (1, 2), (2, 1), O.ACT_RELU)]}[pad]
    for sem, st, dil, act in combos:
        spec = O.ConvSpec(2, 6, 7, cin, 3, 3, cout, 1, st[0], st[1], dil[0], dil[1], padding, pad_values, act, sem)
        names = _run_all_dst_mfma(spec, seed=cin * 3 + cout, tile=tile)
        assert all("bconv2d_mfma" in n and ",%dx%d>" % tile in n for n in names), names


def test_mfma_engine_reference_grid_sample():
    n = 0
    for case in CASES[:72:12] + CASES[72::160]:
        for sem in (O.SEM_REFERENCE, O.SEM_OPTIMIZED):
            if case[2] != 1 or not legal(case[0], case[1], case[2], case[5], sem):
                continue
            spec = _spec(case, sem)
            if spec.out_h <= 0 or spec.out_w <= 0:
                continue
            _run_all_dst_mfma(spec, zlib.crc32(_id(case).encode()) & 0xFFFF)
            n += 1
    assert n >= 10


def test_mfma_engine_batch_chunking_and_pointwise():
    spec = O.ConvSpec(5, 6, 7, 64, 1, 1, 32)
    _run_all_dst_mfma(spec, 11, max_batch=2)
    spec = O.ConvSpec(3, 6, 7, 128, 3, 3, 48, padding=O.PADDING_SAME, pad_values=1, activation=O.ACT_RELU6)
    _run_all_dst_mfma(spec, 12, max_batch=1)


@pytest.mark.parametrize("cin,cout", [(64, 64), (64, 32), (128, 128), (256, 256), (32, 64), (96, 96), (40, 160), (200, 64), (256, 32),
                                      (512, 64), (480, 128), (449, 32), (192, 64), (160, 96), (130, 32)])     # (129..192: the fourth K-step is empty)
@pytest.mark.parametrize("act", [O.ACT_NONE, O.ACT_RELU], ids=["none", "relu"])
def test_pointwise_streaming_kernel(cin, cout, act):
    """The 1x1 streaming kernel (lce_kernels_pointwise.h; filter bank in registers, waves walking 32-pixel tiles
    of the batch's pixel matrix) against the oracle: 1 / 2 / 4 / 8 K-steps incl. partial last words and an empty
    upper K-half, 1 / 2 / 4 channel tiles per block with several blocks along the channels, a pixel count that is
    not a multiple of 32, more tiles than waves (the tile loop and its prefetch), batch chunking."""
    t = cout // 32
    widest = 4 if t % 4 == 0 else 2 if t % 2 == 0 else 1
    if cin > 256:
        widest = min(widest, 2)
    try:
        for k, (b, h, w_, mb) in enumerate(((3, 5, 7, 0), (2, 9, 11, 0), (5, 4, 4, 2))):
            for padding in (O.PADDING_VALID, O.PADDING_SAME):
                # the launch-size rule gives these small launches at most 64 channels per block: ask for the widest and for 32 too
                H.set_pointwise(widest if (k + padding) % 2 == 0 else 1 if k == 1 else 0)
                spec = O.ConvSpec(b, h, w_, cin, 1, 1, cout, padding=padding, pad_values=1, activation=act)
                names = _run_all_dst_mfma(spec, seed=cin + 3 * cout + b, max_batch=mb, engine="pointwise")
                assert all(n.startswith("bconv2d_pointwise<") for n in names), names
                want = widest if (k + padding) % 2 == 0 else 1 if k == 1 else min(widest, 2)
                for n in names:      # (float output: at most 64 channels per block)
                    assert "N%dx32" % (min(want, 2) if "<f32" in n else want) in n, (names, want)
    finally:
        H.set_pointwise(0)


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 256), (200, 96), (512, 128), (192, 128)])
@pytest.mark.parametrize("stride", [(2, 2), (1, 2), (3, 2)], ids=lambda s: "s%dx%d" % s)
def test_pointwise_kernel_strided(cin, cout, stride):
    """Strided 1x1 layers (the shortcut convolutions of ResNet-style binary nets): output pixel (b, oy, ox) reads input
    pixel (b, oy * sh, ox * sw); odd and even extents, VALID and SAME (neither pads a 1x1 filter), batch chunking."""
    try:
        for k, (b, h, w_, mb) in enumerate(((3, 5, 7, 0), (2, 10, 12, 0), (4, 9, 4, 1))):
            for padding in (O.PADDING_VALID, O.PADDING_SAME):
                H.set_pointwise(0 if k else (2 if (cout // 32) % 2 == 0 else 1))
                spec = O.ConvSpec(b, h, w_, cin, 1, 1, cout, 1, stride[0], stride[1], padding=padding, pad_values=1,
                                  activation=O.ACT_RELU if k == 1 else O.ACT_NONE)
                names = _run_all_dst_mfma(spec, seed=cin + cout + b + stride[0], max_batch=mb, engine="pointwise")
                assert all(n.startswith("bconv2d_pointwise<") for n in names), names
    finally:
        H.set_pointwise(0)


def test_pointwise_kernel_refuses_what_it_cannot_run():
    x, w, mul, bias = synth.conv_inputs(O.ConvSpec(1, 4, 4, 64, 3, 3, 64, padding=O.PADDING_SAME, pad_values=1), 1)
    for spec in (O.ConvSpec(1, 4, 4, 64, 3, 3, 64, padding=O.PADDING_SAME, pad_values=1),       # 3x3
                 O.ConvSpec(1, 4, 4, 128, 1, 1, 64, groups=2),                                  # grouped
                 O.ConvSpec(1, 4, 4, 64, 1, 1, 48),                                             # channels not a multiple of 32
                 O.ConvSpec(1, 4, 4, 320, 1, 1, 64),                                            # 5 K-steps: no instance
                 O.ConvSpec(1, 4, 4, 576, 1, 1, 64)):                                           # filter bank too deep
        x, w, mul, bias = synth.conv_inputs(spec, 1)
        with pytest.raises(RuntimeError, match="pointwise kernel runs 1x1"):
            H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="pointwise")


@pytest.mark.parametrize("tile", [(128, 64), (128, 128), (256, 64), (128, 256)], ids=lambda t: "%dx%d" % t)
@pytest.mark.parametrize("cin,cout,pad", [(64, 64, "ONE"), (96, 33, "SAME"), (20, 7, "VALID"), (160, 70, "ONE")])
def test_mfma_direct_variant(tile, cin, cout, pad):
    """The direct variant (input halo expanded into LDS by each block, no FP4 workspace):
    several tiles per image with a partial last one, tiles that start mid-row, strides,
    dilation, both zero-padding semantics, channel counts that are not multiples of 64."""
    padding, pad_values = PADS[pad]
    combos = {"ONE": [(O.SEM_REFERENCE, (1, 1), (1, 1), O.ACT_NONE)],
              "SAME": [(O.SEM_OPTIMIZED, (2, 1), (1, 2), O.ACT_NONE), (O.SEM_REFERENCE, (1, 1), (1, 1), O.ACT_RELU)],
              "VALID": [(O.SEM_REFERENCE, (1, 2), (2, 1), O.ACT_RELU)]}[pad]
    for sem, st, dil, act in combos:
        spec = O.ConvSpec(2, 15, 13, cin, 3, 3, cout, 1, st[0], st[1], dil[0], dil[1], padding, pad_values, act, sem)
        names = _run_all_dst_mfma(spec, seed=cin * 5 + cout, tile=tile, engine="direct")
        assert all("bconv2d_mfma_direct" in n and ",%dx%d>" % tile in n for n in names), names


def test_mfma_direct_variant_shapes():
    # 1x1 filter, 5x5 filter with asymmetric SAME padding (even input, stride 2), batch chunking
    _run_all_dst_mfma(O.ConvSpec(3, 9, 17, 64, 1, 1, 32), 21, tile=(128, 64), engine="direct", max_batch=2)
    _run_all_dst_mfma(O.ConvSpec(1, 12, 14, 32, 5, 5, 40, 1, 2, 2, 1, 1, O.PADDING_SAME, 0, O.ACT_NONE, O.SEM_REFERENCE), 22,
                      tile=(128, 64), engine="direct")
    _run_all_dst_mfma(O.ConvSpec(1, 12, 14, 32, 5, 5, 40, 1, 2, 2, 1, 1, O.PADDING_SAME, 0, O.ACT_NONE, O.SEM_OPTIMIZED), 23,
                      tile=(128, 64), engine="direct")


def test_mfma_direct_variant_small_images_share_a_tile():
    """Images of at most BM/2 pixels: a tile holds several whole images (each with its own
    LDS halo); the batch need not be a multiple of that count."""
    for spec, tile in [
        (O.ConvSpec(5, 7, 7, 64, 3, 3, 40, padding=O.PADDING_SAME, pad_values=1), (128, 64)),           # 2 images / tile
        (O.ConvSpec(7, 5, 4, 96, 3, 3, 33, padding=O.PADDING_SAME, pad_values=0,
                    semantics=O.SEM_REFERENCE), (128, 64)),                                              # 6 images / tile
        (O.ConvSpec(7, 5, 4, 96, 3, 3, 33, padding=O.PADDING_SAME, pad_values=0,
                    semantics=O.SEM_OPTIMIZED), (128, 64)),
        (O.ConvSpec(4, 9, 8, 32, 3, 2, 70, 1, 2, 1, 1, 2, O.PADDING_VALID, 0, O.ACT_RELU), (256, 128)),  # strided, dilated
        (O.ConvSpec(3, 6, 6, 64, 1, 1, 64), (128, 128)),                                                 # pointwise
    ]:
        names = _run_all_dst_mfma(spec, 31 + spec.batch, tile=tile, engine="direct")
        assert all("bconv2d_mfma_direct" in n for n in names), names
    _run_all_dst_mfma(O.ConvSpec(9, 5, 4, 64, 3, 3, 32, padding=O.PADDING_SAME, pad_values=1), 40, tile=(128, 64),
                      engine="direct", max_batch=4)   # launches of 4, 4, 1 images


def test_mfma_int8_wide_epilogue():
    """int8 output with 16-byte row stores (WN tiles transposed together): taken when
    channels_out % 16 == 0 and the block's LDS already holds waves*WN*4 KiB of scratch; ragged
    channel tiles (48 of 128) and ragged pixel tiles included; both variants."""
    scale, zp = synth.int8_quant_params(77)
    for spec, engine, tile in [
        (O.ConvSpec(1, 13, 12, 256, 3, 3, 48, padding=O.PADDING_SAME, pad_values=1, activation=O.ACT_RELU), "direct", (128, 128)),
        (O.ConvSpec(1, 13, 12, 128, 3, 3, 272, padding=O.PADDING_SAME, pad_values=1), "direct", (128, 256)),
        (O.ConvSpec(3, 9, 11, 64, 3, 3, 80), "mfma", (128, 256)),
        (O.ConvSpec(3, 9, 11, 64, 1, 1, 64), "mfma", (256, 128)),
    ]:
        x, w, mul, bias = synth.conv_inputs(spec, 78 + spec.channels_out, negative_mul_fraction=0.2)
        want = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=float(scale), out_zero_point=zp)
        got, name = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=float(scale), out_zero_point=zp,
                              tile=tile, engine=engine)
        assert np.array_equal(got, want), name


@pytest.mark.parametrize("seed", range(3))
def test_mfma_random_specs(seed):
    """A small randomized sweep through both matrix-core variants under the hostile LDS-DMA
    emulation (the GPU suite runs the large one): odd sizes, 1..4-wide filters, strides and
    dilations up to 3, ragged channel counts."""
    g = synth.rng(5000 + seed)
    done = 0
    while done < 3:
        cin = int(g.choice([3, 20, 32, 64, 70, 96, 160]))
        cout = int(g.choice([1, 7, 16, 33, 40, 64]))
        kh, kw = int(g.integers(1, 5)), int(g.integers(1, 5))
        sh, sw, dh, dw = (int(v) for v in g.integers(1, 4, 4))
        pad = str(g.choice(["VALID", "SAME", "ONE"]))
        sem = int(g.choice([O.SEM_REFERENCE, O.SEM_OPTIMIZED]))
        act = int(g.choice([O.ACT_NONE, O.ACT_RELU]))
        padding, pv = PADS[pad]
        if pad == "SAME":
            if sem == O.SEM_REFERENCE and cin % 2:
                padding, pv = PADS["ONE"]
            elif sem == O.SEM_OPTIMIZED:
                act = O.ACT_NONE
        h = int(g.integers((kh - 1) * dh + 1, (kh - 1) * dh + 12))
        w = int(g.integers((kw - 1) * dw + 1, (kw - 1) * dw + 12))
        spec = O.ConvSpec(int(g.integers(1, 4)), h, w, cin, kh, kw, cout, 1, sh, sw, dh, dw, padding, pv, act, sem)
        if spec.out_h <= 0 or spec.out_w <= 0:
            continue
        for engine in ("mfma", "direct"):
            _run_all_dst_mfma(spec, 6000 + seed * 10 + done, tile=(128, 64), engine=engine)
        done += 1


@pytest.mark.parametrize("seed", range(48))
def test_planners_own_choice_on_random_layers(seed):
    """Round 5: whatever kernel the planner's cost estimate picks (streaming with its own or interleaved segments, weight-streaming,
    pointwise, block GEMM, xor-popcount), on random layers of the kind the streaming family serves -- 3x3 and 1x1 filters, 64 .. 512
    input channels (ragged ones included), batches 1 .. 6, 1 .. 256 'compute units' -- all three output types equal the oracle's, with
    the planner's batch chunking and with a forced one."""
    g = synth.rng(9100 + seed)
    k = int(g.choice([3, 3, 3, 1]))
    cin = int(g.choice([64, 128, 256, 512, 96, 200, 320, 40]))
    cout = int(g.choice([16, 32, 48, 64, 80, 128, 192, 256, 304, 512]))
    sh, sw = (int(v) for v in g.choice([1, 1, 2], 2))
    pad = str(g.choice(["ONE", "ONE", "SAME", "VALID"])) if k == 3 else "VALID"
    act = int(g.choice([O.ACT_NONE, O.ACT_RELU, O.ACT_RELU6, O.ACT_RELU_N1_TO_1]))
    padding, pv = PADS[pad]
    if pad == "SAME" and cin % 2:
        padding, pv = PADS["ONE"]
    big = seed % 4 == 3           # every fourth layer: launches of many block steps (interleaved runs, the pointwise kernel, several rounds)
    h, w_ = (int(g.integers(12, 40)), int(g.integers(12, 40))) if big else (int(g.integers(3, 16)), int(g.integers(3, 16)))
    batch = int(g.integers(4, 20)) if big else int(g.integers(1, 7))
    spec = O.ConvSpec(batch, h, w_, cin, k, k, cout, 1, sh, sw, 1, 1, padding, pv, act, O.SEM_REFERENCE)
    cus = int(g.choice([4, 8, 16])) if big else int(g.choice([1, 2, 3, 4, 8, 256]))
    H.set_stream(cus, 0)
    try:
        names = _run_all_dst_mfma(spec, 9200 + seed, engine="auto")
        names += _run_all_dst_mfma(spec, 9300 + seed, max_batch=int(g.integers(1, batch + 1)), engine="auto")
        assert names
    finally:
        H.set_stream(256, 0)


@pytest.mark.parametrize("engine", ["mfma", "direct"])
@pytest.mark.parametrize("cin,cout,groups,tile", [(128, 128, 2, (128, 64)), (192, 256, 2, (128, 128)), (128, 256, 4, (128, 64)),
                                                  (320, 128, 2, (256, 64)), (64, 128, 2, (128, 64))])
def test_mfma_engine_grouped(cin, cout, groups, tile, engine):
    """Grouped convolutions on the matrix cores: a block's channels lie in one group, its K loop covers the
    64-channel chunks that group's input slice touches -- slices that start mid-chunk (Cin/G = 32, 96, 160)
    share a chunk with their neighbour, whose channels carry zero weights."""
    for pad, st in (("ONE", (1, 1)), ("VALID", (2, 1)), ("SAME", (1, 1)))[:3 if engine == "direct" else 2]:
        padding, pad_values = PADS[pad]
        spec = O.ConvSpec(2, 9, 11, cin, 3, 3, cout, groups, st[0], st[1], 1, 1, padding, pad_values,
                          O.ACT_RELU if pad == "VALID" else O.ACT_NONE, O.SEM_REFERENCE)
        names = _run_all_dst_mfma(spec, seed=cin + cout + groups, tile=tile, engine=engine)
        assert all("bconv2d_mfma" in n and ",%dx%d>" % tile in n for n in names), names


@pytest.mark.parametrize("engine,tile", [("direct", (128, 128)), ("direct", (128, 256)), ("mfma", (256, 128)), ("direct", (256, 64))])
def test_mfma_second_output_is_the_lcequantize_of_the_float_output(engine, tile):
    """lce_hip_bconv2d_run_dual: the float epilogue also writes sign(y) bits, word for word what LceQuantize
    makes of the float tensor -- negative multipliers, a RELU clamp (y >= 0 wherever the multiplier is
    positive), partial last tiles and a channel count that leaves padding bits in the last word."""
    for cout, act in ((128, O.ACT_NONE), (136, O.ACT_RELU)):
        spec = O.ConvSpec(2, 11, 13, 64, 3, 3, cout, padding=O.PADDING_SAME, pad_values=1, activation=act)
        x, w, mul, bias = synth.conv_inputs(spec, 77 + cout, negative_mul_fraction=0.3)
        bias = (bias - 40.0 * np.abs(mul)).astype(np.float32)          # both signs occur
        words = np.full(spec.output_shape(O.DST_BITPACKED), 0x5A5A5A5A, np.int32)
        got, name = H.bconv2d(spec, O.DST_F32, x, w, mul, bias, tile=tile, engine=engine, sign_words=words)
        want = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
        assert np.array_equal(got.view(np.int32), want.view(np.int32)), name
        assert np.array_equal(words, O.bitpack(want)), name
        assert 0.02 < ((words.view(np.uint32)[..., 0] & 1) == 1).mean() < 0.98


@pytest.mark.parametrize("engine,tile,k", [("direct", (128, 256), 3), ("direct", (128, 128), 3), ("mfma", (256, 128), 3),
                                          ("pointwise", (0, 0), 1)])
@pytest.mark.parametrize("zp", [-128, -127, -3, 0, 1, 5, 127])
def test_second_output_of_an_int8_layer_is_its_lcequantize(engine, tile, k, zp):
    """lce_hip_bconv2d_run_dual on an int8 plan: bit = (q < out_zero_point), word for word what LceQuantize makes of
    the int8 tensor (quantization.cc:76-114) -- computed from the value BEFORE the rounding against the planner's
    threshold (int8_below_threshold).  The scale is chosen so that results land on and around the zero point,
    ties included (multipliers are multiples of 1/2)."""
    cout = 128
    spec = O.ConvSpec(2, 9, 11, 64, k, k, cout, padding=O.PADDING_SAME, pad_values=1, activation=O.ACT_NONE)
    x, w, mul, bias = synth.conv_inputs(spec, 31 + zp, negative_mul_fraction=0.3)
    mul = (np.sign(mul) * 0.5).astype(np.float32)
    bias = np.zeros_like(bias)
    scale = float(k * k * 4)                        # |y| up to ~ K/scale = 8 around zp; halves occur -> exact ties
    words = np.full(spec.output_shape(O.DST_BITPACKED), 0x5A5A5A5A, np.int32)
    got, name = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=scale, out_zero_point=zp, tile=tile, engine=engine,
                          sign_words=words)
    want = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=scale, out_zero_point=zp)
    assert np.array_equal(got, want), name
    assert np.array_equal(words, O.bitpack(want, zp)), name
    if -127 < zp < 127:
        assert 0.02 < (want < zp).mean() < 0.98 and (want == zp).any()


@pytest.mark.parametrize("engine", ["direct", "mfma"])
@pytest.mark.parametrize("cout,zp", [(80, 12), (48, 1), (80, -5), (112, 127), (16, 40)])
def test_second_output_of_an_int8_layer_with_a_ragged_channel_count(engine, cout, zp):
    """Round 6 (found by the randomized GPU test): the padding bits of a row's last word are 0 (bitpack.h:238-244) whatever the zero
    point -- the block GEMM compares EVERY lane with one threshold, and a padded lane (multiplier and bias 0: value 0) was "below"
    every positive zero point.  Cout = 80: the third word holds 16 channels; the planner now gives padded lanes the bias +inf."""
    spec = O.ConvSpec(1, 5, 6, 20, 2, 2, cout, padding=O.PADDING_SAME, pad_values=1)
    x, w, mul, bias = synth.conv_inputs(spec, 1, negative_mul_fraction=0.2)
    words = np.full(spec.output_shape(O.DST_BITPACKED), 0x5A5A5A5A, np.int32)
    got, name = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=1.0 / 9.0, out_zero_point=zp, engine=engine, sign_words=words)
    want = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=1.0 / 9.0, out_zero_point=zp)
    assert np.array_equal(got, want), name
    assert np.array_equal(words, O.bitpack(want, zp)), name
    assert cout % 32 == 0 or not (words[..., -1].view(np.uint32) >> np.uint32(cout % 32)).any()


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 128), (96, 32)])
def test_pointwise_second_output_of_a_float_layer(cin, cout):
    spec = O.ConvSpec(3, 7, 9, cin, 1, 1, cout, activation=O.ACT_RELU)
    x, w, mul, bias = synth.conv_inputs(spec, cin + cout, negative_mul_fraction=0.3)
    bias = (bias - 10.0 * np.abs(mul)).astype(np.float32)
    words = np.full(spec.output_shape(O.DST_BITPACKED), 0x5A5A5A5A, np.int32)
    got, name = H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="pointwise", sign_words=words)
    want = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
    assert name.startswith("bconv2d_pointwise<"), name
    assert np.array_equal(got.view(np.int32), want.view(np.int32)), name
    assert np.array_equal(words, O.bitpack(want)), name


def test_int8_below_threshold_against_the_rounding():
    """The planner's threshold T(zp): for EVERY float c in [-129, 128) sampled densely around each half-integer,
    (round_sat_i8(c) < zp) == (c < T) -- round_sat_i8 being the kernels' own expression (checked against
    saturate(roundf) for every float by test_round_sat_i8_every_float)."""
    l = H.lib()
    l.hostsim_int8_below_threshold.restype = C.c_float
    for zp in (-200, -128, -127, -64, -1, 0, 1, 2, 63, 126, 127, 128, 300):
        t = np.float32(l.hostsim_int8_below_threshold(C.c_int32(zp)))
        centers = np.arange(-129.0, 128.5, 0.5, dtype=np.float32)
        c = np.concatenate([centers, np.nextafter(centers, np.float32(np.inf)), np.nextafter(centers, np.float32(-np.inf)),
                            np.float32([-0.0, 0.0, 1e-30, -1e-30, 126.99999, -127.99999])]).astype(np.float32)
        sat = np.clip(c, np.float32(-128), np.float32(127))
        q = np.trunc(sat + np.copysign(np.float32(0.49999997), sat)).astype(np.int32)
        assert np.array_equal(q, np.clip(np.sign(c) * np.floor(np.abs(c.astype(np.float64)) + 0.5), -128, 127).astype(np.int32))
        assert np.array_equal(q < zp, c < t), zp


@pytest.mark.parametrize("shape", [(1, 8, 64, 64, 64, 3, 1, "ONE"), (2, 12, 96, 64, 96, 3, 1, "SAME"), (1, 16, 128, 128, 64, 3, 2, "ONE"),
                                   (1, 10, 128, 64, 33, 3, 1, "VALID"), (1, 8, 64, 64, 64, 5, 1, "ONE"), (1, 8, 96, 256, 64, 1, 1, "VALID")],
                         ids=lambda s: "x".join(map(str, s)))
def test_mfma_direct_variant_2d_tiles_on_wide_images(shape):
    """Wide images: the direct variant tiles the output in BM/32 rows x 32 columns instead of row-major strips (the halo is
    the tile's own neighbourhood, not whole image rows).  Partial tiles at the right and bottom edges, strides, a 5x5
    filter, exact SAME-zero padding, all three output types, and the float layer's second output."""
    b, h, w_, cin, cout, k, st, pad = shape
    padding, pv = PADS[pad]
    spec = O.ConvSpec(b, h, w_, cin, k, k, cout, 1, st, st, 1, 1, padding, pv, O.ACT_RELU if pad == "VALID" else O.ACT_NONE,
                      O.SEM_REFERENCE)
    names = _run_all_dst_mfma(spec, seed=h + w_ + cin, engine="direct")
    assert all(n.startswith("bconv2d_mfma_direct<") and n.endswith("/2d") for n in names), names
    x, w, mul, bias = synth.conv_inputs(spec, h, negative_mul_fraction=0.3)
    bias = (bias - 0.45 * spec.filter_h * spec.filter_w * cin * np.abs(mul)).astype(np.float32)
    words = np.full(spec.output_shape(O.DST_BITPACKED), 0x5A5A5A5A, np.int32)
    got, name = H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="direct", sign_words=words)
    want = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
    assert np.array_equal(got.view(np.int32), want.view(np.int32)), name
    if cout % 4 == 0:
        assert np.array_equal(words, O.bitpack(want)), name


def test_direct_variant_keeps_strips_where_2d_tiles_would_pad():
    for h, w_ in ((56, 56), (28, 28), (30, 224), (112, 112)):
        spec = O.ConvSpec(1, h, w_, 64, 3, 3, 64, padding=O.PADDING_SAME, pad_values=1)
        x, w, mul, bias = synth.conv_inputs(spec, 1)
        if h * w_ > 4000:
            continue   # (planner rule only; the big ones are not run here)
        _, name = H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="direct")
        assert not name.endswith("/2d"), name


def test_mfma_engine_refuses_grouped():
    spec = O.ConvSpec(1, 6, 6, 128, 3, 3, 64, groups=2)
    x, w, mul, bias = synth.conv_inputs(spec, 1)
    with pytest.raises(RuntimeError, match="matrix-core engine cannot run"):
        H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="mfma")


# ------------------------------------------------------------------------------------ streaming kernel (engine=stream)
STREAM_SHAPES = [
    # batch, h, w, cin, cout, stride, pad, act, compute units, rows per segment (0 = auto)
    (2, 8, 9, 128, 64, (1, 1), "ONE", O.ACT_NONE, 2, 0),        # two pixel phases (64 channels: four pixel blocks per step)
    (3, 12, 10, 256, 256, (1, 1), "ONE", O.ACT_NONE, 2, 0),     # the BASELINE filter bank; two segments per block
    (3, 12, 10, 256, 256, (1, 1), "ONE", O.ACT_RELU, 3, 4),     # forced 4-row segments: halos re-expanded, ragged blocks
    (2, 7, 7, 64, 128, (1, 1), "ONE", O.ACT_NONE, 1, 0),        # one block walks the whole batch; 49 pixels: partial blocks
    (5, 8, 8, 128, 128, (1, 1), "ONE", O.ACT_RELU6, 2, 0),
    (2, 9, 11, 96, 80, (1, 1), "SAME", O.ACT_NONE, 2, 0),       # exact SAME-zero, 96 channels (a partial last word plane)
    (2, 9, 11, 40, 48, (1, 1), "VALID", O.ACT_RELU, 1, 0),
    (2, 11, 9, 200, 304, (2, 2), "ONE", O.ACT_NONE, 2, 0),      # strides; 304 channels: two channel groups (grid.y)
    (3, 10, 12, 256, 192, (1, 2), "ONE", O.ACT_NONE, 2, 5),
    (2, 13, 37, 64, 64, (2, 1), "VALID", O.ACT_RELU6, 3, 0),
    (2, 10, 9, 192, 96, (1, 1), "ONE", O.ACT_NONE, 2, 0),       # 192 input channels on the 256-channel instance (the fourth chunk: codes 0)
    (2, 9, 10, 320, 128, (1, 1), "ONE", O.ACT_RELU, 2, 0),      # 320 (and 384, 448) on the 512-channel K-split instance
    (3, 7, 7, 384, 64, (1, 1), "SAME", O.ACT_NONE, 1, 0),       # ... with the exact SAME-zero border, flat blocks
    (2, 8, 8, 448, 192, (2, 2), "ONE", O.ACT_NONE, 2, 0),
]


@pytest.mark.parametrize("shape", STREAM_SHAPES, ids=lambda s: "%dx%dx%d_%d-%d_cu%d" % (s[0], s[1], s[2], s[3], s[4], s[8]))
def test_stream_kernel(shape):
    """The weight-stationary streaming kernel (lce_kernels_stream.h) with the planner's tables, against the oracle:
    all three output types, segments / blocks / pixel phases / channel groups, ragged last pixel blocks, strides,
    both padding semantics it supports, the general (partial word) expansion path."""
    b, h, w_, cin, cout, st, pad, act, cus, rows = shape
    padding, pad_values = PADS[pad]
    spec = O.ConvSpec(b, h, w_, cin, 3, 3, cout, 1, st[0], st[1], 1, 1, padding, pad_values, act, O.SEM_REFERENCE)
    H.set_stream(cus, rows)
    try:
        for mb in (0, 2):
            names = _run_all_dst_mfma(spec, seed=cin + 7 * cout + b, max_batch=mb, engine="stream")
            assert all(n.startswith("bconv2d_stream<") for n in names), names
            if rows:
                assert all(",rows%d>" % rows in n for n in names), names
    finally:
        H.set_stream(256, 0)


INTERLEAVED_SHAPES = [
    # batch, h, w, cin, cout, stride, pad, act, compute units, rows per segment, strip width (-1: whole rows), batch chunks
    (3, 12, 10, 256, 256, (1, 1), "ONE", O.ACT_NONE, 2, 4, -1, (0, 2)),    # 9 segments over 2 blocks: runs of 5 and 4, ragged pixel blocks
    (5, 8, 8, 128, 128, (1, 1), "ONE", O.ACT_RELU6, 3, 4, -1, (0, 3)),     # two pixel phases; 10 segments over 3 blocks (4 / 3 / 3)
    (4, 8, 8, 64, 64, (1, 1), "SAME", O.ACT_NONE, 3, 2, -1, (0,)),         # four pixel phases, exact SAME-zero, 16 segments over 3 blocks
    (3, 11, 9, 200, 304, (2, 2), "ONE", O.ACT_NONE, 4, 3, -1, (0, 2)),     # strides; two channel groups (grid.y: 2 blocks in x), 6 segments
    (2, 14, 7, 512, 128, (1, 1), "ONE", O.ACT_NONE, 3, 7, -1, (0,)),       # K split over wave pairs, 4 segments over 3 blocks
    (2, 4, 64, 256, 192, (1, 1), "ONE", O.ACT_NONE, 3, 2, 32, (0,)),       # column strips: 8 segments (image, strip, rows) over 3 blocks
]


@pytest.mark.parametrize("shape", INTERLEAVED_SHAPES, ids=lambda s: "%dx%dx%d_%d-%d_cu%d_rows%d" % (s[0], s[1], s[2], s[3], s[4], s[8], s[9]))
def test_stream_kernel_interleaved_runs(shape):
    """Round 5: block b of the streaming kernel owns segments b, b + grid, b + 2 grid, ... instead of consecutive ones (the
    launch writes one compact window that moves through the output).  Same bytes as the oracle for all three output types,
    uneven runs, smaller last chunks of a batch (which keep the planned stride), strips, the K-split kernel."""
    b, h, w_, cin, cout, st, pad, act, cus, rows, wso, chunks = shape
    padding, pad_values = PADS[pad]
    spec = O.ConvSpec(b, h, w_, cin, 3, 3, cout, 1, st[0], st[1], 1, 1, padding, pad_values, act, O.SEM_REFERENCE)
    H.set_stream(cus, rows)
    H.set_stream_strip(wso)
    H.set_stream_interleave(1)
    try:
        for mb in chunks:
            names = _run_all_dst_mfma(spec, seed=cin + 5 * cout + b, max_batch=mb, engine="stream")
            assert all(n.startswith("bconv2d_stream<") for n in names), names
            if mb == 0:     # (a chunk so small that every block owns ONE segment has nothing to interleave)
                assert all(",il>" in n or ",il," in n for n in names), names      # (",il,x2>": two blocks per CU, bitpacked 64-channel bank)
    finally:
        H.set_stream(256, 0)
        H.set_stream_strip(-1)
        H.set_stream_interleave(0)


@pytest.mark.parametrize("shape", [
    # batch, h, w, cin, cout, stride, pad, compute units
    (5, 12, 10, 64, 64, (1, 1), "ONE", 2),      # 4 blocks on 2 CUs, four pixel phases, ragged pixel blocks
    (3, 9, 16, 40, 96, (1, 1), "SAME", 3),      # general expansion path (40 channels), exact SAME-zero, two channel slices x two phases
    (4, 13, 11, 64, 200, (2, 2), "ONE", 2),     # strides; four slices, a ragged last word
    (1, 8, 8, 33, 32, (1, 1), "VALID", 4),      # one image over 8 blocks
], ids=lambda s: "%dx%dx%d_%d-%d_cu%d" % (s[0], s[1], s[2], s[3], s[4], s[7]))
def test_stream_kernel_two_blocks_per_cu(shape):
    """Round 6: the bitpacked-output instance of the 64-input-channel bank is compiled for two resident blocks per CU (229 registers,
    no epilogue scratch) and the planner launches 2 x CUs blocks where both blocks' LDS fit (lce_plan_stream.cpp).  Same bytes as the
    oracle with one and with two blocks per CU and with the estimate's own choice; the other instances refuse the option."""
    b, h, w_, cin, cout, st, pad, cus = shape
    padding, pad_values = PADS[pad]
    spec = O.ConvSpec(b, h, w_, cin, 3, 3, cout, 1, st[0], st[1], 1, 1, padding, pad_values, O.ACT_NONE, O.SEM_REFERENCE)
    x, w, mul, bias = synth.conv_inputs(spec, 7 * cin + cout, negative_mul_fraction=0.2)
    thr = O.thresholds_converter(spec, mul, bias)
    thr[::5] = np.iinfo(np.int32).max
    thr[1::7] = np.iinfo(np.int32).min
    want = O.bconv2d(spec, O.DST_BITPACKED, x, w, thresholds=thr)
    H.set_stream(cus, 0)
    try:
        for occ, mb in ((2, 0), (2, 2), (1, 0), (0, 0), (0, 3)):      # (mb: launches of at most mb images -- a smaller last chunk)
            H.set_stream_blocks_per_cu(occ)
            got, name = H.bconv2d(spec, O.DST_BITPACKED, x, w, thresholds=thr, engine="stream", max_batch=mb)
            assert np.array_equal(got, want), (name, mb)
            assert name.startswith("bconv2d_stream<bitpacked,3x3x64,") and (",x2>" in name) == (occ == 2 or (occ == 0 and ",x2>" in name)), name
            if occ == 1:
                assert ",x2" not in name, name
        H.set_stream_blocks_per_cu(2)
        with pytest.raises(RuntimeError, match="stream_blocks_per_cu=2"):
            H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="stream")
    finally:
        H.set_stream(256, 0)
        H.set_stream_blocks_per_cu(0)


@pytest.mark.parametrize("phases,cout,cus", [(2, 256, 2), (4, 256, 4)])
def test_stream_kernel_with_forced_pixel_phases(phases, cout, cus):
    """stream_pixel_phases: a block's four waves as 2 slices x 2 pixel blocks (or 1 x 4) also when there are >= 3 channel
    slices -- half (a quarter) of the filter bank per block, grid.y picks the slices.  Same results, all output types;
    the first block step takes its weights as they arrive (round 4), whatever the mapping."""
    spec = O.ConvSpec(3, 6, 8, 256, 3, 3, cout, padding=O.PADDING_SAME, pad_values=1)
    H.set_stream(cus, 0)
    H.set_stream_phases(phases)
    try:
        names = _run_all_dst_mfma(spec, seed=11 + phases + cout, max_batch=0, engine="stream")
        assert all(n.startswith("bconv2d_stream<") and ",phases%d>" % phases in n for n in names), names
    finally:
        H.set_stream(256, 0)
        H.set_stream_phases(0)


KSPLIT_SHAPES = [
    # batch, h, w, cin, cout, stride, padding, activation, compute units, batch chunks to try
    (3, 7, 7, 512, 128, (1, 1), "ONE", O.ACT_NONE, 1, (0, 2)),     # QuickNet's last section: 49-pixel images, blocks cut across images; a short last run
    (3, 6, 5, 512, 192, (1, 1), "ONE", O.ACT_RELU, 2, (0,)),       # 30-pixel images, three slices: an idle wave pair in the second grid row
    (1, 9, 11, 480, 64, (1, 1), "SAME", O.ACT_NONE, 1, (0,)),      # a partial last chunk, exact SAME-zero, one slice: the second pair of the block idles
    (2, 8, 6, 512, 128, (2, 2), "VALID", O.ACT_RELU6, 1, (0,)),    # strides, VALID
]


@pytest.mark.parametrize("shape", KSPLIT_SHAPES, ids=lambda s: "%dx%dx%d_%d-%d_cu%d" % (s[0], s[1], s[2], s[3], s[4], s[8]))
def test_stream_kernel_k_split_over_wave_pairs(shape):
    """512 input channels on the streaming kernel (round 4): the K dimension split over a pair of waves, partial sums
    swapped through LDS before a halved epilogue; pixel blocks cut from the block's images laid end to end where an
    image does not fill 32-pixel blocks.  All three output types against the oracle, with batch chunking."""
    b, h, w_, cin, cout, st, pad, act, cus, chunks = shape
    padding, pad_values = PADS[pad]
    spec = O.ConvSpec(b, h, w_, cin, 3, 3, cout, 1, st[0], st[1], 1, 1, padding, pad_values, act, O.SEM_REFERENCE)
    H.set_stream(cus, 0)
    try:
        for mb in chunks:
            names = _run_all_dst_mfma(spec, seed=cin + 3 * cout + b, max_batch=mb, engine="stream")
            assert all(n.startswith("bconv2d_stream<") and "3x3x512" in n for n in names), names
    finally:
        H.set_stream(256, 0)


@pytest.mark.parametrize("cin,cout", [(512, 128)])
def test_stream_flat_pixel_blocks_second_output(cin, cout):
    """Blocks cut across 7x7 images + the second (sign) output + int8: rows of two images in one pixel block."""
    spec = O.ConvSpec(3, 7, 7, cin, 3, 3, cout, padding=O.PADDING_SAME, pad_values=1, activation=O.ACT_NONE)
    x, w, mul, bias = synth.conv_inputs(spec, 5 + cin, negative_mul_fraction=0.3)
    bias = (bias - np.median(O.bconv2d(spec, O.DST_F32, x, w, mul, bias), axis=(0, 1, 2))).astype(np.float32)
    words = np.full(spec.output_shape(O.DST_BITPACKED), 0x5A5A5A5A, np.int32)
    H.set_stream(1, 0)
    try:
        got, name = H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="stream", sign_words=words)
        want = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
        assert name.startswith("bconv2d_stream<f32"), name
        assert np.array_equal(got.view(np.int32), want.view(np.int32)), name
        assert np.array_equal(words, O.bitpack(want)), name
        words[:] = 0x5A5A5A5A
        got, name = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=8.0, out_zero_point=2, engine="stream", sign_words=words)
        want = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=8.0, out_zero_point=2)
        assert np.array_equal(got, want), name
        assert np.array_equal(words, O.bitpack(want, 2)), name
    finally:
        H.set_stream(256, 0)


STRIP_SHAPES = [
    # batch, h, w, cin, cout, stride, padding, activation, compute units, strip width, batch chunks to try
    (2, 4, 64, 256, 192, (1, 1), "ONE", O.ACT_NONE, 3, 32, (0,)),       # two strips per image; runs that pass from strip to strip and image to image
    (1, 4, 96, 256, 192, (1, 1), "SAME", O.ACT_RELU, 2, 32, (0,)),      # exact SAME-zero: columns AND rows outside the image are zeros
    (1, 5, 128, 200, 192, (1, 2), "ONE", O.ACT_NONE, 1, 32, (0,)),      # column stride 2 (64 outputs), partial word planes (200 channels)
    (1, 8, 128, 256, 128, (2, 1), "ONE", O.ACT_NONE, 2, 64, (0,)),      # 64-wide strips, row stride 2, two pixel phases
    (1, 4, 66, 256, 192, (1, 1), "VALID", O.ACT_NONE, 1, 32, (0,)),     # VALID: 64 outputs from 66 columns
    (1, 16, 128, 256, 256, (2, 1), "VALID", O.ACT_NONE, 4, 64, (0,)),   # 126 outputs: not a multiple of the strip -> refused
]


@pytest.mark.parametrize("shape", STRIP_SHAPES, ids=lambda s: "%dx%dx%d_%d-%d_w%d" % (s[0], s[1], s[2], s[3], s[4], s[9]))
def test_stream_kernel_column_strips(shape):
    """Wide images on the streaming kernel (round 4): segments are runs of rows of ONE column strip, a ring row holds the
    strip + its halo columns, columns outside the image are produced as padding.  Forced on small images here; shapes
    whose output width the strip does not divide are refused (the planner then stays with whole rows / the block GEMM)."""
    b, h, w_, cin, cout, st, pad, act, cus, wso, chunks = shape
    padding, pad_values = PADS[pad]
    spec = O.ConvSpec(b, h, w_, cin, 3, 3, cout, 1, st[0], st[1], 1, 1, padding, pad_values, act, O.SEM_REFERENCE)
    H.set_stream(cus, 0)
    H.set_stream_strip(wso)
    try:
        if spec.out_w % wso:
            x, w, mul, bias = synth.conv_inputs(spec, 1)
            with pytest.raises(RuntimeError, match="stream_strip"):
                H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="stream")
            return
        for mb in chunks:
            names = _run_all_dst_mfma(spec, seed=cin + cout + w_, max_batch=mb, engine="stream")
            assert all(n.startswith("bconv2d_stream<") and ",strips%d>" % wso in n for n in names), names
    finally:
        H.set_stream(256, 0)
        H.set_stream_strip(-1)


WSTREAM_SHAPES = [
    # batch, h, w, cin, cout, stride, pad, act, compute units, batch chunks to try
    (3, 14, 14, 256, 256, (1, 1), "ONE", O.ACT_NONE, 4, (0, 2)),      # QuickNet's 14x14x256: one image per group, parts of 4 + 3 pixel blocks, a partial last block
    (5, 7, 7, 512, 512, (1, 1), "ONE", O.ACT_RELU, 4, (0, 3)),        # 7x7x512: several images per group, flat pixel blocks, two blocks in y, a short last group
    (2, 9, 11, 200, 304, (1, 1), "SAME", O.ACT_NONE, 2, (0,)),        # exact SAME-zero, partial word planes (200 channels), 304 channels: a short last slice group
    (3, 11, 9, 128, 192, (2, 2), "ONE", O.ACT_RELU6, 2, (0, 1)),      # strides, 128 input channels, three of four waves active
    (2, 10, 12, 256, 192, (1, 2), "VALID", O.ACT_NONE, 3, (0,)),      # VALID padding, column stride 2
    (1, 4, 4, 512, 256, (1, 1), "ONE", O.ACT_NONE, 8, (0,)),          # a single half-empty pixel block
    (2, 4, 4, 128, 16, (1, 1), "SAME", O.ACT_RELU, 2, (0,)),          # sixteen output channels: one wave, a quarter of its slice
    (2, 9, 8, 192, 128, (1, 1), "ONE", O.ACT_NONE, 2, (0,)),          # 192 input channels on the 256-channel instance
    (2, 7, 7, 320, 256, (1, 1), "ONE", O.ACT_RELU, 2, (0, 1)),        # 320 on the 512-channel one
]


@pytest.mark.parametrize("shape", WSTREAM_SHAPES, ids=lambda s: "%dx%dx%d_%d-%d_cu%d" % (s[0], s[1], s[2], s[3], s[4], s[8]))
def test_wstream_kernel(shape):
    """The weight-streaming kernel (lce_kernels_wstream.h, round 5: activations stationary in LDS, weights streamed into registers
    during the K loop) with the planner's tables, against the oracle: all three output types, groups of several images cut into
    flat 32-pixel blocks, parts of unequal length, partial last blocks and short last groups, strides, both padding semantics."""
    b, h, w_, cin, cout, st, pad, act, cus, chunks = shape
    padding, pad_values = PADS[pad]
    spec = O.ConvSpec(b, h, w_, cin, 3, 3, cout, 1, st[0], st[1], 1, 1, padding, pad_values, act, O.SEM_REFERENCE)
    H.set_stream(cus, 0)
    try:
        for mb in chunks:
            names = _run_all_dst_mfma(spec, seed=cin + 3 * cout + b, max_batch=mb, engine="wstream")
            assert all(n.startswith("bconv2d_wstream<") for n in names), names
    finally:
        H.set_stream(256, 0)


@pytest.mark.parametrize("cin,cout,hw,zp", [(256, 256, 14, 3), (512, 320, 7, -5)])
def test_wstream_second_output_is_the_lcequantize_of_the_output(cin, cout, hw, zp):
    """run_dual on the weight-streaming kernel: the sign words are the LceQuantize of the float / int8 values it stores."""
    spec = O.ConvSpec(3, hw, hw, cin, 3, 3, cout, padding=O.PADDING_SAME, pad_values=1, activation=O.ACT_NONE)
    x, w, mul, bias = synth.conv_inputs(spec, 9 + cin, negative_mul_fraction=0.3)
    bias = (bias - np.median(O.bconv2d(spec, O.DST_F32, x, w, mul, bias), axis=(0, 1, 2))).astype(np.float32)
    words = np.full(spec.output_shape(O.DST_BITPACKED), 0x5A5A5A5A, np.int32)
    H.set_stream(4, 0)
    try:
        got, name = H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="wstream", sign_words=words)
        want = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
        assert name.startswith("bconv2d_wstream<f32"), name
        assert np.array_equal(got.view(np.int32), want.view(np.int32)), name
        assert np.array_equal(words, O.bitpack(want)), name
        words[:] = 0x5A5A5A5A
        got, name = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=8.0, out_zero_point=zp, engine="wstream", sign_words=words)
        want = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=8.0, out_zero_point=zp)
        assert np.array_equal(got, want), name
        assert np.array_equal(words, O.bitpack(want, zp)), name
    finally:
        H.set_stream(256, 0)


@pytest.mark.parametrize("engine,shape", [("stream", (2, 8, 8, 64, 64, 3)), ("stream", (2, 6, 6, 256, 64, 3)), ("wstream", (2, 7, 7, 256, 64, 3)),
                                          ("pointwise", (2, 9, 9, 128, 64, 1))], ids=lambda v: v if isinstance(v, str) else "x".join(map(str, v)))
def test_int8_floor_rounding_is_taken_only_where_it_is_exact(engine, shape):
    """Round 5: the streaming / weight-streaming / pointwise kernels round int8 outputs with floor(y + 0.5) (one instruction) on plans where
    that provably equals the reference's round-half-away -- no accumulator value inside the clamps lands on an exact negative tie for any
    channel -- and with the exact sequence otherwise.  Random parameters: floor instances, equal to the oracle.  Multipliers of -0.25 with
    zero bias: y = -x / 4 hits -k - 0.5 whenever x = 2 (mod 4): the planner must keep the exact instances, and the bytes still equal the
    oracle's (with floor rounding every such tie would be off by one)."""
    b, h, w_, cin, cout, k = shape
    spec = O.ConvSpec(b, h, w_, cin, k, k, cout, padding=O.PADDING_SAME if k == 3 else O.PADDING_VALID, pad_values=1 if k == 3 else 0)
    x, w, mul, bias = synth.conv_inputs(spec, 17 + cin)
    H.set_stream(2, 0)
    try:
        got, name = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=0.5, out_zero_point=-7, engine=engine)
        assert H.last_int8_floor() == 1, name
        assert np.array_equal(got, O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=0.5, out_zero_point=-7)), name
        mul_t = np.full(cout, -0.25, np.float32)
        mul_t[::3] = 0.25
        bias_t = np.zeros(cout, np.float32)
        got, name = H.bconv2d(spec, O.DST_I8, x, w, mul_t, bias_t, out_scale=1.0, out_zero_point=0, engine=engine)
        want = O.bconv2d(spec, O.DST_I8, x, w, mul_t, bias_t, out_scale=1.0, out_zero_point=0)
        assert H.last_int8_floor() == 0, name
        assert np.array_equal(got, want), name
    finally:
        H.set_stream(256, 0)


def test_int8_floor_rounding_with_an_adjusted_bias_equals_the_oracle():
    """A layer with real-valued parameters does hold an exact negative tie now and then (about 2^-17 per value near |y| = 100); the
    planner lowers that channel's bias by a few 2^-17 and proves the whole channel again.  Parameters are drawn until a plan with an
    adjusted channel shows up (about every second draw of this shape); every plan's bytes equal the oracle's."""
    spec = O.ConvSpec(1, 5, 5, 256, 3, 3, 96, padding=O.PADDING_SAME, pad_values=1)
    adjusted = 0
    H.set_stream(2, 0)
    try:
        for seed in range(12):
            rng = np.random.default_rng(900 + seed)
            x, w, _, _ = synth.conv_inputs(spec, seed)
            mul = rng.uniform(0.02, 0.09, 96).astype(np.float32) * rng.choice([-1.0, 1.0], 96).astype(np.float32)
            bias = rng.uniform(-20.0, 20.0, 96).astype(np.float32)
            got, name = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=0.73, out_zero_point=3, engine="stream")
            assert np.array_equal(got, O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=0.73, out_zero_point=3)), (seed, name)
            if H.last_int8_floor() == 1:
                adjusted += H.last_int8_adjusted()
            if adjusted:
                break
        assert adjusted > 0
    finally:
        H.set_stream(256, 0)


def test_int8_floor_proof_leaves_non_finite_parameters_to_the_exact_instances():
    """A NaN parameter or an infinite multiplier (0 * inf: unspecified in the reference) must not be 'proven': the plan keeps the
    round-half-away instances and the original parameters; in every case the finite channels still equal the oracle's bytes."""
    spec = O.ConvSpec(1, 5, 5, 64, 3, 3, 32, padding=O.PADDING_SAME, pad_values=1)
    x, w, mul, bias = synth.conv_inputs(spec, 5)
    want = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=0.37, out_zero_point=2)
    H.set_stream(2, 0)
    try:
        for bad in (np.inf, -np.inf, np.nan):
            for which in ("mul", "bias"):
                m2, b2 = mul.copy(), bias.copy()
                (m2 if which == "mul" else b2)[3] = bad
                got, name = H.bconv2d(spec, O.DST_I8, x, w, m2, b2, out_scale=0.37, out_zero_point=2, engine="stream")
                if which == "mul" or bad != bad:          # (an infinite bias saturates every value: that IS provable)
                    assert H.last_int8_floor() == 0, (bad, which, name)
                keep = np.arange(32) != 3
                assert np.array_equal(got[..., keep], want[..., keep]), (bad, which, name)
    finally:
        H.set_stream(256, 0)


def test_int8_floor_proof_by_bisection_equals_the_full_enumeration(monkeypatch):
    """The planner checks the run of accumulator values whose output lies inside the clamps (found by bisection: y is monotone in x), one
    value either side and the two ends; LCE_PLAN_INT8_FULL=1 checks every value.  Same verdict and same number of adjusted channels on
    parameters of every kind: tiny and large multipliers, both signs, zero, activations that cut into int8's range, ties all over."""
    rng = np.random.default_rng(77)
    verdicts = []
    for case in range(24):
        cin, cout = [(64, 32), (128, 48), (256, 64)][case % 3]
        act = [O.ACT_NONE, O.ACT_RELU, O.ACT_RELU6, O.ACT_RELU_N1_TO_1][case % 4]
        spec = O.ConvSpec(1, 4, 4, cin, 3, 3, cout, padding=O.PADDING_SAME, pad_values=1, activation=act)
        x, w, _, _ = synth.conv_inputs(spec, case)
        kind = case % 6
        mul = rng.uniform(0.001, [0.05, 0.5, 4.0][kind % 3], cout).astype(np.float32) * rng.choice([-1.0, 1.0], cout).astype(np.float32)
        if kind == 3:
            mul = np.where(rng.random(cout) < 0.5, 0.25, -0.125).astype(np.float32)
        if kind == 4:
            mul[::5] = 0.0
        bias = (rng.uniform(-30, 30, cout) if kind != 3 else rng.integers(-9, 9, cout)).astype(np.float32)
        scale, zp = float(rng.choice([0.0625, 0.37, 1.0, 2.5])), int(rng.integers(-20, 21))
        got = []
        for full in ("", "1"):
            if full:
                monkeypatch.setenv("LCE_PLAN_INT8_FULL", "1")
            else:
                monkeypatch.delenv("LCE_PLAN_INT8_FULL", raising=False)
            out, name = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=scale, out_zero_point=zp, engine="stream")
            assert np.array_equal(out, O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=scale, out_zero_point=zp)), (case, name, full)
            got.append((H.last_int8_floor(), H.last_int8_adjusted()))
        assert got[0] == got[1], (case, got)
        verdicts.append(got[0])
    assert any(v[0] == 1 for v in verdicts) and any(v[0] == 0 for v in verdicts) and any(v[1] > 0 for v in verdicts), verdicts


def test_wstream_kernel_refuses_what_it_cannot_run():
    for spec, why in [
        (O.ConvSpec(1, 6, 6, 64, 3, 3, 64, padding=O.PADDING_SAME, pad_values=1), "65 .. 512 input channels"),
        (O.ConvSpec(1, 6, 6, 256, 1, 1, 64), "3x3"),
        (O.ConvSpec(1, 120, 120, 256, 3, 3, 64, padding=O.PADDING_SAME, pad_values=1), "does not fit"),     # one image: 122 x 122 x 144 B
    ]:
        x, w, mul, bias = synth.conv_inputs(spec, 2)
        with pytest.raises(RuntimeError, match=why):
            H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="wstream")


def test_stream_kernel_refuses_what_it_cannot_run():
    x, w, mul, bias = synth.conv_inputs(O.ConvSpec(1, 6, 6, 64, 3, 3, 64), 1)
    for spec, why in [
        (O.ConvSpec(1, 6, 6, 64, 3, 3, 70, padding=O.PADDING_SAME, pad_values=1), "16-byte groups"),      # float: Cout % 4
        (O.ConvSpec(1, 6, 6, 64, 1, 1, 64), "3x3"),
        (O.ConvSpec(1, 6, 6, 576, 3, 3, 64, padding=O.PADDING_SAME, pad_values=1), "at most 512 input"),
        (O.ConvSpec(1, 8, 8, 64, 3, 3, 64, 1, 1, 1, 2, 2, O.PADDING_SAME, 1), "dilation"),
        (O.ConvSpec(1, 6, 6, 128, 3, 3, 128, 2, padding=O.PADDING_SAME, pad_values=1), "ungrouped"),
    ]:
        x, w, mul, bias = synth.conv_inputs(spec, 2)
        with pytest.raises(RuntimeError, match="streaming kernel"):
            H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="stream")
    # stream_rows must divide the output height
    spec = O.ConvSpec(2, 9, 9, 64, 3, 3, 64, padding=O.PADDING_SAME, pad_values=1)
    x, w, mul, bias = synth.conv_inputs(spec, 3)
    H.set_stream(2, 4)
    try:
        with pytest.raises(RuntimeError, match="divide the output height"):
            H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="stream")
    finally:
        H.set_stream(256, 0)


def test_stream_kernel_is_the_auto_choice_for_a_256_channel_3x3_layer_that_fills_the_chip():
    spec = O.ConvSpec(6, 12, 16, 256, 3, 3, 256, padding=O.PADDING_SAME, pad_values=1)
    x, w, mul, bias = synth.conv_inputs(spec, 5)
    want = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
    H.set_stream(4, 0)          # a 4-CU "device": 6 images fill it, 12 block steps per block
    try:
        got, name = H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="auto")
    finally:
        H.set_stream(256, 0)
    assert name.startswith("bconv2d_stream<") and np.array_equal(got.view(np.int32), want.view(np.int32)), name
    # 256 CUs: round 4's rule sent a launch that fills 6 of them to the block GEMM; the cost estimate (round 5) cuts the six images
    # into 2-row segments over 36 blocks of the streaming kernel (profiles/r05/engine_sweep_box*.jsonl: the faster choice at small batch)
    got, name = H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="auto")
    assert name.startswith("bconv2d_stream<") and np.array_equal(got.view(np.int32), want.view(np.int32)), name


@pytest.mark.parametrize("cin,cout,act,cus", [(256, 256, O.ACT_NONE, 2), (128, 136, O.ACT_RELU, 1), (64, 64, O.ACT_NONE, 3)])
def test_stream_second_output_is_the_lcequantize_of_the_float_output(cin, cout, act, cus):
    """lce_hip_bconv2d_run_dual on the streaming kernel: the woven epilogue also writes sign(y) bits, word for word what
    LceQuantize makes of the float tensor -- negative multipliers, a RELU clamp, ragged last pixel blocks (143 pixels),
    a channel count that leaves padding bits (136), pixel phases (64 channels)."""
    spec = O.ConvSpec(3, 11, 13, cin, 3, 3, cout, padding=O.PADDING_SAME, pad_values=1, activation=act)
    x, w, mul, bias = synth.conv_inputs(spec, 77 + cout, negative_mul_fraction=0.3)
    bias = (bias - np.median(O.bconv2d(spec, O.DST_F32, x, w, mul, bias), axis=(0, 1, 2))).astype(np.float32)   # both signs occur
    words = np.full(spec.output_shape(O.DST_BITPACKED), 0x5A5A5A5A, np.int32)
    H.set_stream(cus, 0)
    try:
        got, name = H.bconv2d(spec, O.DST_F32, x, w, mul, bias, engine="stream", sign_words=words)
    finally:
        H.set_stream(256, 0)
    want = O.bconv2d(spec, O.DST_F32, x, w, mul, bias)
    assert name.startswith("bconv2d_stream<f32"), name
    assert np.array_equal(got.view(np.int32), want.view(np.int32)), name
    assert np.array_equal(words, O.bitpack(want)), name
    assert 0.02 < ((words.view(np.uint32)[..., 0] & 1) == 1).mean() < 0.98


@pytest.mark.parametrize("zp", [-128, -127, -3, 0, 1, 5, 127])
def test_stream_second_output_of_an_int8_layer_is_its_lcequantize(zp):
    cout = 144 if zp == 5 else 128          # 144: half a word of padding bits, which must stay 0
    spec = O.ConvSpec(2, 9, 11, 64, 3, 3, cout, padding=O.PADDING_SAME, pad_values=1, activation=O.ACT_NONE)
    x, w, mul, bias = synth.conv_inputs(spec, 31 + zp, negative_mul_fraction=0.3)
    mul = (np.sign(mul) * 0.5).astype(np.float32)
    bias = np.zeros_like(bias)
    scale = float(3 * 3 * 4)                        # |y| up to ~ K/scale = 8 around zp; halves occur -> exact ties
    words = np.full(spec.output_shape(O.DST_BITPACKED), 0x5A5A5A5A, np.int32)
    H.set_stream(2, 0)
    try:
        got, name = H.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=scale, out_zero_point=zp, engine="stream", sign_words=words)
    finally:
        H.set_stream(256, 0)
    want = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=scale, out_zero_point=zp)
    assert name.startswith("bconv2d_stream<i8"), name
    assert np.array_equal(got, want), name
    assert np.array_equal(words, O.bitpack(want, zp)), name
