"""GPU parity tests added in round 6 (review items 4 and the advisor's round-5 findings), through the C ABI against the oracle:

* `LceDequantize` against the oracle's `unpack_matrix` restatement, including the saturating branches and scales whose
  reciprocal rounds (tflite/kernels/quantization.cc:131-138);
* int8 plans whose one-instruction epilogue forms run on ADJUSTED per-channel parameters (csrc/lce_plan.cpp,
  prepare_int8_epilogue), on inputs that produce EVERY accumulator value the plan can reach, for the streaming kernel, the
  weight-streaming kernel, the pointwise kernel and the pointwise plan's fallback to the block GEMM;
* >= 400 randomized (layer, batch, output type, run / run_dual) draws where the planner's cost estimate decides, each on
  `engine=auto` AND on every engine that accepts the shape: bytes equal to the oracle's and equal across engines."""
import os

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, seed, settings
from hypothesis import strategies as st

import oracle_lib as O
import synth
from lce_amd import amd

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
NTHREADS = os.cpu_count() or 8


def _params(spec, dst, **kw):
    return amd.ConvParams(spec.batch, spec.in_h, spec.in_w, spec.channels_in, spec.filter_h, spec.filter_w, spec.channels_out,
                          spec.groups, spec.stride_h, spec.stride_w, spec.dilation_h, spec.dilation_w, spec.padding,
                          spec.pad_values, spec.activation, dst, spec.semantics, **kw)


# ------------------------------------------------------------------------------------ LceDequantize vs the oracle

@pytest.mark.parametrize("zp,scale", [(120, 0.1), (-125, 1.0 / 7.0), (3, 0.3), (0, 1.0), (-128, 0.5), (127, 0.02), (-7, 0.4),
                                      (10, 2.5), (-20, 1.0 / 3.0), (90, 0.025)])
def test_lcedequantize_int8_against_the_oracle(zp, scale):
    """quantization.cc:131-138: offset = TfLiteRound(1 / scale); a 0 bit -> min(127, zp + offset), a 1 bit -> max(-128, zp - offset).
    The round-5 test only made round trips with scale = 1 / n and |zp| <= 20, which reach neither saturating branch nor a
    reciprocal that rounds ((120, 0.1): 130 -> 127; (-125, 1/7): -132 -> -128; 0.3 -> 3.33 -> 3; 0.4 -> 2.5 (in float: just
    below) ; 2.5 -> 0.4 -> 0: both bits give the zero point).  Flat (whole words, aligned) and row kernels, ragged columns."""
    g = synth.rng(int(zp) * 31 + int(scale * 1000))
    for shape, cols in (((3, 5, 7, 1), 1), ((2, 4, 4, 1), 31), ((2, 9, 3, 1), 32), ((1, 6, 6, 2), 33), ((4, 7, 7, 2), 64),
                        ((2, 5, 5, 4), 100), ((8, 14, 14, 8), 256)):
        words = synth.random_words(g, shape)             # (padding bits arbitrary: the op reads the first `cols` bits of a row)
        want = O.unpack(words, cols, np.int8, scale=float(np.float32(scale)), zero_point=zp)
        got = amd.unpack(torch.from_numpy(words).to(DEV), cols, torch.int8, scale=float(np.float32(scale)), zero_point=zp)
        assert np.array_equal(got.cpu().numpy(), want), (shape, cols)
    # the two values really are the saturated ones where the case says so
    offset = int(np.floor(np.float32(1.0) / np.float32(scale) + np.float32(0.5)))
    vals = set(np.unique(want).tolist())
    assert vals <= {min(127, zp + offset), max(-128, zp - offset)}


@pytest.mark.parametrize("cols", [1, 31, 32, 33, 64, 100, 256])
def test_lcedequantize_float_and_bool_against_the_oracle(cols):
    g = synth.rng(500 + cols)
    words = synth.random_words(g, (3, 6, 5, (cols + 31) // 32))
    for dt, tdt in ((np.float32, torch.float32), (np.bool_, torch.bool)):
        want = O.unpack(words, cols, dt)
        got = amd.unpack(torch.from_numpy(words).to(DEV), cols, tdt).cpu().numpy()
        assert np.array_equal(got, want), (cols, dt)


# ------------------------------------------------------------------------------------ int8: adjusted parameters x every accumulator value

def _every_accumulator_inputs(spec, g):
    """Inputs on which every output channel of `spec` (ONE filter shared by all channels, stride = filter extent: patches do not
    overlap) sees accumulator value k at output pixel k (mod K + 1), k = 0 .. K = KH*KW*Cin: pixel p's patch is the filter with
    exactly p bits flipped (core/types.h:45-47: the accumulator is popcount(a ^ w))."""
    kh, kw, cin, cw = spec.filter_h, spec.filter_w, spec.channels_in, spec.in_words
    k_total = kh * kw * cin
    filt = synth.random_words(g, (1, kh, kw, cw), cin)
    w = np.repeat(filt, spec.channels_out, axis=0)
    fbits = ((filt.view(np.uint32)[0, :, :, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(kh, kw, cw * 32)[:, :, :cin]
    x = np.zeros((spec.batch, spec.in_h, spec.in_w, cw), np.uint32)
    p = 0
    for b in range(spec.batch):
        for oy in range(spec.out_h):
            for ox in range(spec.out_w):
                flips = p % (k_total + 1)
                mask = np.zeros(k_total, np.uint8)
                mask[g.permutation(k_total)[:flips]] = 1
                patch = fbits ^ mask.reshape(kh, kw, cin)
                padded = np.zeros((kh, kw, cw * 32), np.uint32)
                padded[:, :, :cin] = patch
                words = (padded.reshape(kh, kw, cw, 32) << np.arange(32, dtype=np.uint32)).sum(axis=-1, dtype=np.uint64).astype(np.uint32)
                x[b, oy * kh:(oy + 1) * kh, ox * kw:(ox + 1) * kw, :] = words
                p += 1
    assert p > k_total, "the image must hold every accumulator value"
    return x.view(np.int32), w


@pytest.mark.parametrize("engine,kernel,shape", [
    ("stream", "bconv2d_stream<i8", (2, 54, 54, 64, 3, 256)),         # K = 576: 577 values, 2 x 18 x 18 = 648 output pixels
    ("wstream", "bconv2d_wstream<i8", (19, 24, 24, 128, 3, 256)),     # K = 1152: 1153 values, 19 x 8 x 8 = 1216 pixels (whole images in LDS)
    ("pointwise", "bconv2d_pointwise<i8", (2, 12, 12, 256, 1, 256)),  # K = 256: 257 values, 288 pixels
    ("pointwise-unaligned", "bconv2d_pointwise<i8", (2, 12, 12, 256, 1, 256)),   # ... the plan's block-GEMM fallback (output not 16-byte aligned)
])
def test_int8_adjusted_parameters_over_every_reachable_accumulator(engine, kernel, shape):
    """The advisor's round-5 finding: the proof behind the one-instruction int8 forms enumerates the accumulator values on the HOST
    (std::fmaf, volatile floats) and then ships per-channel parameters that differ from the folded ones by up to 2 ulps / 4 grid
    steps.  Here the DEVICE runs such plans (int8_epilogue() reports adjusted channels) on inputs that produce every accumulator
    value 0 .. K in every channel, and the bytes must be the oracle's on the ORIGINAL parameters -- for the streaming kernel
    (v_fma + v_cvt_rpi), the weight-streaming kernel (v_pk_fma + v_cvt_rpi), the pointwise kernel (two roundings + v_cvt_rpi) and
    the pointwise plan's fallback to the block GEMM, which sees the adjusted parameters with ITS arithmetic (two roundings,
    round-half-away)."""
    b, h, w_, cin, k, cout = shape
    unaligned = engine.endswith("unaligned")
    eng = engine.split("-")[0]
    spec = O.ConvSpec(b, h, w_, cin, k, k, cout, stride_h=k, stride_w=k, padding=O.PADDING_VALID, activation=O.ACT_NONE)
    adjusted_plans = 0
    k_total = k * k * cin
    for seed in range(24):
        g = np.random.default_rng(4200 + seed)
        scale, zp = float(g.choice([0.21, 0.73, 0.125, 1.0])), int(g.integers(-20, 21))
        # y swings over 120 .. 250 int8 steps while the accumulator goes 0 .. 2K: most of a channel's values land inside int8's range,
        # where |y| ~ 100 lives on a 2^-17 grid and exact ties / near-ties do occur (a plan with an adjusted channel: every second to
        # sixth draw, checked on the host -- int8_epilogue() needs no GPU)
        span = g.uniform(120.0, 250.0, cout)
        mul = (span / (2 * k_total) * scale * g.choice([-1.0, 1.0], cout)).astype(np.float32)
        bias = g.uniform(-20.0, 20.0, cout).astype(np.float32)
        probe_w = synth.random_words(g, (1, k, k, spec.in_words), cin)
        plan = amd.Bconv2dPlan(_params(spec, amd.I8, out_scale=scale, out_zero_point=zp))
        plan.set_weights(np.repeat(probe_w, cout, axis=0), mul, bias)
        plan.set_option("engine", eng)
        forms, adjusted = plan.int8_epilogue()
        assert plan.kernel_name().startswith(kernel), plan.kernel_name()
        if not (forms and adjusted > 0):
            continue
        adjusted_plans += 1
        x, w = _every_accumulator_inputs(spec, g)            # (built only for the plans that are run: ~1 s of NumPy each)
        plan.set_weights(w, mul, bias)
        assert plan.int8_epilogue() == (forms, adjusted)     # (the proof depends on the parameters, not on the filter bits)
        want = O.bconv2d(spec, O.DST_I8, x, w, mul, bias, out_scale=scale, out_zero_point=zp, threads=NTHREADS)
        xd = torch.from_numpy(x).to(DEV)
        if unaligned:
            n = int(np.prod(plan.output_shape))
            buf = torch.full((n + 64,), 0x5A, dtype=torch.int8, device=DEV)
            plan.run_ptr(xd.data_ptr(), buf.data_ptr() + 3, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = buf[3:3 + n].reshape(plan.output_shape).cpu().numpy()
            assert int(buf[2]) == 0x5A and int(buf[3 + n]) == 0x5A
        else:
            got = plan.run(xd).cpu().numpy()
        assert np.array_equal(got, want), (seed, plan.kernel_name(), adjusted)
        # the same plan with the reference's own sequence on the ORIGINAL parameters
        plan.set_option("int8_rounding", "exact")
        assert plan.int8_epilogue() == (False, 0)
        assert np.array_equal(plan.run(xd).cpu().numpy(), want), (seed, "exact")
        if adjusted_plans >= 3:
            break
    assert adjusted_plans >= 1, "no draw produced a plan with adjusted channels: the case does not test what it claims"


# ------------------------------------------------------------------------------------ where the planner decides: randomized draws

BATCHES = [1, 3, 16, 64, 201, 256]
# ("stream_x2": engine=stream with two resident blocks per CU -- refused by every instance but the one compiled for it)
ENGINES = ["stream", "stream_x2", "wstream", "direct", "mfma", "pointwise", "valu"]


@st.composite
def _layer(draw):
    """(spec, dst, dual): the families whose kernel the planner's cost estimate picks -- 3x3 layers of 64..512 input channels
    (weight-stationary / weight-streaming / block GEMM), 1x1 layers (pointwise / block GEMM) -- plus odd shapes that fall to the
    general paths.  Work is bounded so that the oracle (all host cores) and the six engines take well under a second a draw."""
    family = draw(st.sampled_from(["3x3", "3x3", "3x3", "1x1", "odd", "grid"]))
    batch = draw(st.sampled_from(BATCHES))
    dst = draw(st.sampled_from([O.DST_F32, O.DST_I8, O.DST_BITPACKED]))
    dual = dst != O.DST_BITPACKED and draw(st.booleans())
    act = draw(st.sampled_from([O.ACT_NONE, O.ACT_NONE, O.ACT_RELU, O.ACT_RELU6]))
    if family == "3x3":
        cin = draw(st.sampled_from([64, 128, 192, 256, 320, 512]))
        cout = draw(st.sampled_from([32, 64, 96, 128, 256]))
        hw = draw(st.sampled_from([(4, 4), (7, 7), (8, 12), (14, 14), (9, 30)]))
        stride = draw(st.sampled_from([1, 1, 2]))
        pad = draw(st.sampled_from([(O.PADDING_SAME, 1), (O.PADDING_SAME, 1), (O.PADDING_VALID, 0), (O.PADDING_SAME, 0)]))
        k = 3
    elif family == "1x1":
        cin = draw(st.sampled_from([64, 128, 160, 256, 512]))
        cout = draw(st.sampled_from([32, 64, 128, 256]))
        hw = draw(st.sampled_from([(7, 7), (14, 14), (10, 13)]))
        stride = draw(st.sampled_from([1, 1, 2]))
        pad = (O.PADDING_VALID, 0)
        k = 1
    elif family == "grid":
        # the reference's op-test grid widened (tflite/tests/bconv2d_test.cc:790-856): groups, dilation, rectangular filters and
        # strides, both SAME-zero semantics, ragged channel counts -- with the second output and every engine on top
        groups = draw(st.sampled_from([1, 1, 2, 4]))
        cin = draw(st.sampled_from([32, 64, 96, 128, 160])) * groups if groups > 1 else draw(st.sampled_from([3, 20, 64, 70, 96, 192, 300]))
        cout = draw(st.sampled_from([1, 7, 16, 33, 40, 64, 100, 136])) * groups
        kh, kw = draw(st.integers(1, 4)), draw(st.integers(1, 4))
        sh, sw = draw(st.integers(1, 3)), draw(st.integers(1, 3))
        dh, dw = draw(st.integers(1, 2)), draw(st.integers(1, 3))
        h = (kh - 1) * dh + 1 + draw(st.integers(0, 12))
        w_ = (kw - 1) * dw + 1 + draw(st.integers(0, 12))
        padname = draw(st.sampled_from(["VALID", "SAME0", "ONE"]))
        sem = draw(st.sampled_from([O.SEM_REFERENCE, O.SEM_OPTIMIZED]))
        pad = {"VALID": (O.PADDING_VALID, 0), "SAME0": (O.PADDING_SAME, 0), "ONE": (O.PADDING_SAME, 1)}[padname]
        if padname == "SAME0":   # bconv2d.cc:188-200: what Prepare accepts
            if sem == O.SEM_REFERENCE and (cin // groups) % 2:
                pad = (O.PADDING_SAME, 1)
            elif sem == O.SEM_OPTIMIZED:
                act, dst, dual = O.ACT_NONE, O.DST_F32, dual and dst == O.DST_F32
        batch = min(batch, 16)
        spec = O.ConvSpec(batch, h, w_, cin, kh, kw, cout, groups, sh, sw, dh, dw, pad[0], pad[1], act, sem)
        return spec, dst, dual, draw(st.integers(0, 10_000))
    else:
        cin = draw(st.sampled_from([20, 96, 200]))
        cout = draw(st.sampled_from([7, 33, 80]))
        hw = draw(st.sampled_from([(5, 6), (9, 9)]))
        stride = draw(st.sampled_from([1, 2]))
        pad = draw(st.sampled_from([(O.PADDING_SAME, 1), (O.PADDING_VALID, 0)]))
        k = draw(st.sampled_from([2, 3]))
    # bound the work of one draw (binary MACs): big batches get the small maps
    while batch * hw[0] * hw[1] * cout * k * k * cin > 6.0e10 and batch > 1:
        batch = BATCHES[BATCHES.index(batch) - 1]
    sem = O.SEM_REFERENCE
    if pad == (O.PADDING_SAME, 0) and cin % 2:
        pad = (O.PADDING_SAME, 1)
    spec = O.ConvSpec(batch, hw[0], hw[1], cin, k, k, cout, 1, stride, stride, 1, 1, pad[0], pad[1], act, sem)
    seed = draw(st.integers(0, 10_000))
    return spec, dst, dual, seed


def _run_engine(spec, dst, dual, engine, x, w, mul, bias, thr, scale, zp):
    adst = {O.DST_F32: amd.F32, O.DST_I8: amd.I8, O.DST_BITPACKED: amd.BITPACKED}[dst]
    plan = amd.Bconv2dPlan(_params(spec, adst, out_scale=float(scale), out_zero_point=int(zp)))
    if dst == O.DST_BITPACKED:
        plan.set_weights(w, None, None, thr)
    else:
        plan.set_weights(w, mul, bias)
    plan.set_option("engine", engine.split("_")[0])
    if engine == "stream_x2":
        plan.set_option("stream_blocks_per_cu", "2")
    if dual:
        y, bits = plan.run_dual(x)
        return y.cpu().numpy(), bits.cpu().numpy(), plan.kernel_name()
    return plan.run(x).cpu().numpy(), None, plan.kernel_name()


_seen_kernels = set()
_draws = [0]


# (a longer hunt on the GPU box: LCE_FUZZ_EXAMPLES=5000 LCE_FUZZ_SEED=1 python -m pytest <this file> -- other draws than the suite's)
_FUZZ_N = int(os.environ.get("LCE_FUZZ_EXAMPLES", "420"))
_FUZZ_SEED = os.environ.get("LCE_FUZZ_SEED")


@(seed(int(_FUZZ_SEED)) if _FUZZ_SEED else (lambda f: f))
@settings(max_examples=_FUZZ_N, deadline=None, derandomize=_FUZZ_SEED is None, database=None,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large, HealthCheck.filter_too_much])
@given(_layer())
def test_where_the_planner_decides(case):
    spec, dst, dual, seed = case
    if spec.out_h <= 0 or spec.out_w <= 0:
        return
    _draws[0] += 1
    x, w, mul, bias = synth.conv_inputs(spec, seed, negative_mul_fraction=0.2)
    scale, zp = synth.int8_quant_params(seed)
    if seed % 7 == 0:           # the ends of int8's range now and then (the second output's threshold sits at them too)
        zp = (-128, 127, -127, 126)[(seed // 7) % 4]
    thr = None
    if dst == O.DST_BITPACKED:
        thr = O.thresholds_converter(spec, mul, bias)
        thr[::5] = np.iinfo(np.int32).max
        thr[1::7] = np.iinfo(np.int32).min
        want = O.bconv2d(spec, dst, x, w, thresholds=thr, threads=NTHREADS)
    else:
        want = O.bconv2d(spec, dst, x, w, mul, bias, out_scale=float(scale), out_zero_point=zp, threads=NTHREADS)
    want_bits = O.bitpack(want, zp if dst == O.DST_I8 else 0) if dual else None
    xd = torch.from_numpy(x).to(DEV)
    ran = []
    for engine in ["auto"] + ENGINES:
        try:
            got, bits, name = _run_engine(spec, dst, dual, engine, xd, w, mul, bias, thr, scale, zp)
        except amd.LceHipError as e:
            assert engine != "auto", e             # (auto always runs; a forced engine may refuse the shape)
            assert e.code in (amd.ERR_UNSUPPORTED, amd.ERR_INVALID), e
            continue
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (engine, name, spec, dst, dual)
        if dual:
            assert np.array_equal(bits, want_bits), (engine, name, spec, dst, "second output")
        ran.append(name)
        _seen_kernels.add(name.split("<")[0] + (",x2" if name.endswith(",x2>") else ""))
    assert ran


def test_where_the_planner_decides_reached_every_family():
    """(runs after the draws above, same process) the 300 draws exercised every kernel family."""
    if _draws[0] == 0:
        pytest.skip("the randomized test did not run in this process")
    assert _draws[0] >= min(350, _FUZZ_N - 70), _draws[0]
    want = {"bconv2d_stream", "bconv2d_stream,x2", "bconv2d_wstream", "bconv2d_mfma_direct", "bconv2d_mfma", "bconv2d_pointwise", "bconv2d_tiled"}
    assert want <= _seen_kernels, sorted(want - _seen_kernels)


# ------------------------------------------------------------------------------------ pointers a host may hand over

@pytest.mark.parametrize("engine,shape", [("stream", (3, 14, 14, 256, 3, 64)), ("wstream", (3, 14, 14, 256, 3, 64)),
                                          ("stream", (2, 20, 20, 64, 3, 32)), ("direct", (3, 14, 14, 256, 3, 64)),
                                          ("pointwise", (2, 12, 12, 128, 1, 64))])
def test_tensors_need_no_16_byte_alignment(engine, shape):
    """An interpreter's arena gives 16- or 64-byte aligned tensors, but nothing in the C ABI asks for it (include/lce_hip.h: "device
    pointers"; the reference's kernels take any pointer).  The streaming kernels store rows as 16-byte pieces through buffer
    resources and load input words 16 bytes at a time: both work on bases that are only element-aligned (int8 output at +1 / +4,
    float and bitpacked output at +4, input words at +4), bytes equal to the oracle's and nothing written outside the tensor."""
    b, h, w_, cin, k, cout = shape
    pad = (O.PADDING_SAME, 1) if k > 1 else (O.PADDING_VALID, 0)
    spec = O.ConvSpec(b, h, w_, cin, k, k, cout, padding=pad[0], pad_values=pad[1])
    x, w, mul, bias = synth.conv_inputs(spec, 5)
    thr = O.thresholds_converter(spec, mul, bias)
    xin = torch.zeros((x.size + 8,), dtype=torch.int32, device=DEV)
    xin[1:1 + x.size] = torch.from_numpy(x.ravel()).to(DEV)                    # the input tensor starts 4 bytes into its buffer
    for dst, odst, off in ((amd.I8, O.DST_I8, 1), (amd.I8, O.DST_I8, 4), (amd.F32, O.DST_F32, 4), (amd.BITPACKED, O.DST_BITPACKED, 4)):
        plan = amd.Bconv2dPlan(_params(spec, dst, out_scale=0.25, out_zero_point=2))
        plan.set_weights(w, mul, bias, thr if dst == amd.BITPACKED else None)
        plan.set_option("engine", engine)
        want = O.bconv2d(spec, odst, x, w, mul, bias, thresholds=thr, out_scale=0.25, out_zero_point=2)
        buf = torch.full((want.nbytes + 64,), 0x5A, dtype=torch.uint8, device=DEV)
        plan.run_ptr(xin.data_ptr() + 4, buf.data_ptr() + off, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = buf.cpu().numpy()
        assert np.array_equal(got[off:off + want.nbytes], want.view(np.uint8).ravel()), (plan.kernel_name(), dst, off)
        assert (got[:off] == 0x5A).all() and (got[off + want.nbytes:] == 0x5A).all(), (plan.kernel_name(), dst, off, "wrote outside")


# ------------------------------------------------------------------------------------ two resident blocks per CU (bitpacked, 64-channel bank)

@pytest.mark.parametrize("shape", [
    # batch, h = w, cin, cout, stride
    (256, 56, 64, 64, 1),      # config 5's first 3x3 layer with bitpacked output: 512 blocks, two per CU
    (40, 56, 64, 64, 1),       # fewer segments than 2 x CUs at whole images: shorter segments
    (9, 112, 40, 96, 1),       # general expansion path, two channel slices x two pixel phases
    (64, 28, 64, 256, 2),      # strides, four slices
])
def test_two_blocks_per_cu_equal_one_block_per_cu_and_the_oracle(shape):
    """Round 6: `stream_blocks_per_cu` (auto / 1 / 2) changes the launch -- twice the blocks, two resident per CU, on the one instance
    compiled for it -- never the bytes.  Full-size launches (the co-resident blocks share a CU's LDS and run their barriers
    independently); the oracle checks a slice of the batch, the rest is compared between the launches."""
    b, hw, cin, cout, stride = shape
    spec = O.ConvSpec(b, hw, hw, cin, 3, 3, cout, 1, stride, stride, padding=O.PADDING_SAME, pad_values=1)
    x, w, mul, bias = synth.conv_inputs(spec, 3 * cin + cout)
    thr = O.thresholds_converter(spec, mul, bias)
    xd = torch.from_numpy(x).to(DEV)
    outs, names = {}, {}
    for occ in ("1", "2", "auto"):
        plan = amd.Bconv2dPlan(_params(spec, amd.BITPACKED))
        plan.set_weights(w, None, None, thr)
        plan.set_option("engine", "stream")
        plan.set_option("stream_blocks_per_cu", occ)
        out = torch.full(plan.output_shape, -7, dtype=torch.int32, device=DEV)
        plan.run(xd, out)
        torch.cuda.synchronize()
        outs[occ], names[occ] = out.cpu().numpy(), plan.kernel_name()
    assert ",x2>" in names["2"] and ",x2" not in names["1"], names
    nb = min(b, 3)
    sub = spec.with_batch(nb)
    want = O.bconv2d(sub, O.DST_BITPACKED, x[:nb], w, thresholds=thr)
    assert np.array_equal(outs["1"][:nb], want), names
    assert np.array_equal(outs["2"], outs["1"]) and np.array_equal(outs["auto"], outs["1"]), names


def test_two_blocks_per_cu_is_refused_where_no_instance_is_compiled_for_it():
    """(the 128-channel bank's bitpacked instance and the 64-channel bank's int8 / float instances need more than 256 registers or
    32 KiB of transpose scratch per block)"""
    for dst, cin in ((amd.BITPACKED, 128), (amd.I8, 64), (amd.F32, 64)):
        s = O.ConvSpec(4, 28, 28, cin, 3, 3, 128, padding=O.PADDING_SAME, pad_values=1)
        x, w, mul, bias = synth.conv_inputs(s, 1)
        plan = amd.Bconv2dPlan(_params(s, dst, out_scale=0.25, out_zero_point=2))
        plan.set_weights(w, mul, bias, O.thresholds_converter(s, mul, bias) if dst == amd.BITPACKED else None)
        plan.set_option("engine", "stream")
        plan.set_option("stream_blocks_per_cu", "2")
        with pytest.raises(amd.LceHipError, match="stream_blocks_per_cu=2"):
            plan.run(torch.from_numpy(x).to(DEV))
