"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/lce_hip.h declares, does Prepare-style validation / shape inference / parameter
folding without a GPU, and FAILS LOUDLY (no fallback) when asked to compute without one."""
import ctypes as C
import itertools
import os
import re

import numpy as np
import pytest

import oracle_lib as O
import synth
from lce_amd import amd
from test_oracle_vs_float_conv import CASES, PADS, legal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "lce_hip.h")).read()
    declared = set(re.findall(r"\b(lce_hip_[a-z0-9_]+)\s*\(", header))
    assert declared == set(amd.ABI_SYMBOLS), declared ^ set(amd.ABI_SYMBOLS)
    lib = amd.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.lce_hip_abi_version() == 3
    # the in-tree library is the product build: no timing ablation / A-B switch compiled in (csrc/lce_experiments.h)
    assert lib.lce_hip_build_flavor() == b"product"


def _params(spec: O.ConvSpec, dst, **kw):
    return amd.ConvParams(spec.batch, spec.in_h, spec.in_w, spec.channels_in, spec.filter_h,
                          spec.filter_w, spec.channels_out, spec.groups, spec.stride_h, spec.stride_w,
                          spec.dilation_h, spec.dilation_w, spec.padding, spec.pad_values,
                          spec.activation, dst, spec.semantics, **kw)


def test_shape_inference_matches_oracle_on_reference_grid():
    seen = 0
    for case in CASES[::5]:
        inp, flt, g, st, dil, pad, act = case
        if not legal(inp, flt, g, pad, O.SEM_REFERENCE):
            continue
        padding, pv = PADS[pad]
        spec = O.ConvSpec(inp[0], inp[1], inp[2], inp[3], flt[0], flt[1], flt[2], g, st[0], st[1],
                          dil[0], dil[1], padding, pv, act, O.SEM_REFERENCE)
        if spec.out_h <= 0 or spec.out_w <= 0:
            with pytest.raises(amd.LceHipError):
                amd.Bconv2dPlan(_params(spec, amd.F32))
            continue
        for dst in (amd.F32, amd.I8, amd.BITPACKED):
            plan = amd.Bconv2dPlan(_params(spec, dst))
            assert plan.output_shape == spec.output_shape(dst)
            assert plan.padding() == (spec.pad_h, spec.pad_w)
            plan.close()
            seen += 1
    assert seen > 100


@pytest.mark.parametrize("dst,act", itertools.product((amd.F32, amd.I8), (0, 1, 2, 3)))
def test_folding_matches_oracle(dst, act):
    """OneTimeSetup (tflite/kernels/bconv2d.cc:324-392), bit for bit."""
    spec = O.ConvSpec(1, 6, 6, 96, 3, 3, 40, padding=O.PADDING_SAME, pad_values=1, activation=act)
    _, filt, mul, bias = synth.conv_inputs(spec, 17 + act, negative_mul_fraction=0.25)
    scale, zp = synth.int8_quant_params(5)
    plan = amd.Bconv2dPlan(_params(spec, dst, out_scale=float(scale), out_zero_point=zp))
    plan.set_weights(filt, mul, bias)
    got = plan.folded()
    want = O.fold_output_transform(spec, dst, mul, bias, float(scale), zp)
    assert np.array_equal(got[0].view(np.int32), want[0].view(np.int32))
    assert np.array_equal(got[1].view(np.int32), want[1].view(np.int32))
    assert got[2:] == want[2:]
    a = spec.backtransform_add          # NONE [0,2a], RELU [0,a], RELU6 [a-6,a], N1_TO_1 [a-1,a+1]
    assert got[2:] == {0: (0, 2 * a), 1: (0, a), 3: (a - 6, a), 2: (a - 1, a + 1)}[act]


def test_prepare_rejections_match_reference_messages():
    """bconv2d.cc:109-112,173-200 and the death tests bconv2d_test.cc:858-917."""
    base = dict(batch=1, in_height=16, in_width=16, channels_in=64, filter_height=3, filter_width=3,
                channels_out=128, padding=amd.PADDING_SAME, pad_values=0)
    for kw in (dict(dst_type=amd.F32, activation=amd.ACT_RELU, semantics=amd.SEM_OPTIMIZED),
               dict(dst_type=amd.BITPACKED, semantics=amd.SEM_OPTIMIZED),
               dict(dst_type=amd.I8, semantics=amd.SEM_OPTIMIZED)):
        with pytest.raises(amd.LceHipError, match="Zero-padding is only supported by"):
            amd.Bconv2dPlan(amd.ConvParams(**base, **kw))
    # the reference registration accepts all three as long as channels_in is even
    for dst in (amd.F32, amd.I8, amd.BITPACKED):
        amd.Bconv2dPlan(amd.ConvParams(**base, dst_type=dst, semantics=amd.SEM_REFERENCE)).close()
    odd = dict(base, channels_in=33)
    with pytest.raises(amd.LceHipError, match="Zero-padding is only supported by"):
        amd.Bconv2dPlan(amd.ConvParams(**odd, semantics=amd.SEM_REFERENCE))
    with pytest.raises(amd.LceHipError, match="pad_values must be 0 or 1"):
        amd.Bconv2dPlan(amd.ConvParams(**dict(base, pad_values=2)))
    with pytest.raises(amd.LceHipError):      # group size must be a multiple of 32 (:180-185)
        amd.Bconv2dPlan(amd.ConvParams(**dict(base, groups=4, pad_values=1)))
    with pytest.raises(amd.LceHipError):      # channels_out % groups (:186)
        amd.Bconv2dPlan(amd.ConvParams(**dict(base, groups=2, channels_out=127, pad_values=1)))


def test_kernel_selection_is_host_side_and_named():
    spec = O.ConvSpec(256, 56, 56, 256, 3, 3, 256, padding=O.PADDING_SAME, pad_values=1)
    _, filt, mul, bias = synth.conv_inputs(O.ConvSpec(1, 3, 3, 256, 3, 3, 256), 1)
    plan = amd.Bconv2dPlan(_params(spec, amd.F32))
    plan.set_weights(filt, mul, bias)
    assert plan.kernel_name() == "bconv2d_stream<f32,3x3x256,rows56>"  # auto: matrix cores, weight-stationary streaming kernel
    plan.set_option("engine", "direct")
    assert plan.kernel_name() == "bconv2d_mfma_direct<f32,128x256>"   # the block GEMM, LDS-halo variant
    plan.set_option("engine", "valu")
    assert plan.kernel_name().startswith("bconv2d_tiled<f32,TM=")     # xor-popcount engine
    plan.set_option("kernel", "general")
    assert plan.kernel_name() == "bconv2d_general<f32>"
    plan.set_option("kernel", "tiled")
    plan.set_option("tile", "1x32")
    assert plan.kernel_name() == "bconv2d_tiled<f32,TM=1,TN=32,CH=4>"
    with pytest.raises(amd.LceHipError):
        plan.set_option("tile", "3x7")
    plan.set_option("kernel", "auto")
    plan.set_option("engine", "mfma")
    plan.set_option("tile", "128x128")
    assert plan.kernel_name() == "bconv2d_mfma<f32,128x128>"         # workspace GEMM variant
    plan.set_option("engine", "direct")
    assert plan.kernel_name() == "bconv2d_mfma_direct<f32,128x128>"
    plan.set_option("tile", "auto")
    plan.set_option("engine", "auto")
    # 512 input channels at batch 256 (round 4): the streaming kernel with the K dimension split over wave pairs, whole 7x7
    # images per segment, pixel blocks cut across the four images of a block
    small = amd.Bconv2dPlan(amd.ConvParams(256, 7, 7, 512, 3, 3, 512, padding=amd.PADDING_SAME, pad_values=1))
    assert small.kernel_name() == "bconv2d_stream<f32,3x3x512,rows7>"
    # ... and on the block GEMM small images share a tile (7x7: two whole images per 128-pixel tile)
    small.set_option("engine", "direct")
    assert small.kernel_name() == "bconv2d_mfma_direct<f32,128x128>"
    # 2048 input channels: the LDS halo of any tile is too big -> workspace GEMM
    deep = amd.Bconv2dPlan(amd.ConvParams(8, 28, 28, 2048, 3, 3, 256, padding=amd.PADDING_SAME, pad_values=1))
    assert deep.kernel_name().startswith("bconv2d_mfma<f32,")
    # 128 input channels, float, batch 256: float rows of a low-K layer are store-bound -- interleaved runs of 4-row segments write one
    # compact window (round 5); the block GEMM when asked for
    mid = amd.Bconv2dPlan(amd.ConvParams(256, 28, 28, 128, 3, 3, 128, padding=amd.PADDING_SAME, pad_values=1))
    assert mid.kernel_name() == "bconv2d_stream<f32,3x3x128,rows4,il>"
    mid.set_option("engine", "direct")
    assert mid.kernel_name() == "bconv2d_mfma_direct<f32,128x128>"
    midb = amd.Bconv2dPlan(amd.ConvParams(256, 28, 28, 128, 3, 3, 128, padding=amd.PADDING_SAME, pad_values=1, dst_type=amd.BITPACKED))
    assert midb.kernel_name() == "bconv2d_stream<bitpacked,3x3x128,rows28>"      # (strided ones too)
    mids = amd.Bconv2dPlan(amd.ConvParams(256, 28, 28, 128, 3, 3, 256, stride_height=2, stride_width=2, padding=amd.PADDING_SAME,
                                          pad_values=1, dst_type=amd.BITPACKED))
    assert mids.kernel_name() == "bconv2d_stream<bitpacked,3x3x128,rows14>"
    grouped = amd.Bconv2dPlan(amd.ConvParams(1, 8, 8, 128, 3, 3, 64, groups=2))
    grouped.set_option("engine", "mfma")
    assert grouped.kernel_name() == ""                              # refused: grouped convolution
    assert "matrix-core engine cannot run" in amd.lib().lce_hip_last_error().decode()


@pytest.mark.skipif(amd.device_count() > 0, reason="this test is about the GPU-less container")
def test_compute_fails_loudly_without_a_gpu():
    """No CPU fallback: the product must not quietly compute on the host."""
    spec = O.ConvSpec(1, 4, 4, 64, 3, 3, 16)
    x, filt, mul, bias = synth.conv_inputs(spec, 1)
    plan = amd.Bconv2dPlan(_params(spec, amd.F32))
    plan.set_weights(filt, mul, bias)
    with pytest.raises(amd.LceHipError) as e:
        plan.run_host(x)
    assert e.value.code == amd.ERR_NO_DEVICE and "no CPU fallback" in e.value.message
    buf = np.zeros(64, np.float32)
    out = np.zeros(2, np.int32)
    rc = amd.lib().lce_hip_bitpack(amd.F32, buf.ctypes.data_as(C.c_void_p), 1, 64, 0,
                                   out.ctypes.data_as(C.c_void_p), None)
    assert rc == amd.ERR_NO_DEVICE


def test_empty_batch_is_legal_and_a_no_op():
    """TFLite tensors may have a zero batch; the reference's loops then do not run
    (core/bconv2d/reference.h:84, core/bmaxpool.h:43).  No GPU is needed for nothing."""
    spec = O.ConvSpec(1, 5, 5, 64, 3, 3, 16)
    _, filt, mul, bias = synth.conv_inputs(spec, 1)
    plan = amd.Bconv2dPlan(amd.ConvParams(0, 5, 5, 64, 3, 3, 16))
    assert plan.output_shape == (0, 3, 3, 16)
    plan.set_weights(filt, mul, bias)
    plan.run_ptr(0, 0)                                   # returns OK without touching anything
    assert amd.lib().lce_hip_bmaxpool(None, 0, 4, 4, 2, 2, 2, 2, 2, amd.PADDING_VALID, None, None) == 0
    with pytest.raises(amd.LceHipError):
        amd.Bconv2dPlan(amd.ConvParams(-1, 5, 5, 64, 3, 3, 16))


@pytest.mark.parametrize("hw,c,dst,want", [
    (56, 256, "F32", "bconv2d_stream<f32,3x3x256,rows56>"),         # BASELINE L0: the weight-stationary streaming kernel,
    (56, 256, "I8", "bconv2d_stream<i8,3x3x256,rows56>"),           # one image per block (round 3; profiles/r03/)
    (14, 256, "I8", "bconv2d_wstream<i8,3x3x256,images1,blocks4>"),   # round 5: weights streamed, activations stationary (store-light single-round layers)
    (14, 256, "BITPACKED", "bconv2d_wstream<bitpacked,3x3x256,images1,blocks4>"),
    (56, 256, "BITPACKED", "bconv2d_stream<bitpacked,3x3x256,rows56>"),
    (56, 64, "F32", "bconv2d_mfma_direct<f32,256x64>"),            # QuickNet stages
    (28, 128, "F32", "bconv2d_stream<f32,3x3x128,rows4,il>"),      # round 5: interleaved runs (profiles/r05/interleaved_runs.txt)
    (56, 64, "I8", "bconv2d_stream<i8,3x3x64,rows56>"),            # int8: with and without the second output
    (28, 128, "BITPACKED", "bconv2d_stream<bitpacked,3x3x128,rows28>"),   # since the ballots lost their padding
    (14, 256, "F32", "bconv2d_stream<f32,3x3x256,rows14>"),
    (7, 512, "F32", "bconv2d_stream<f32,3x3x512,rows7>"),          # round 4: K split over wave pairs, blocks cut across 4 images
    (7, 512, "I8", "bconv2d_stream<i8,3x3x512,rows7>"),
])
def test_planner_choices_for_the_baseline_layers(hw, c, dst, want):
    """The planner prices every candidate kernel (csrc/lce_plan.cpp, estimate_*_us; calibrated on profiles/r05/engine_sweep_box*.jsonl, held to the
    measured best by tests/test_planner_choice.py); this pins what it picks for the BASELINE.json layers at batch 256 so that a planner edit shows
    up as a diff."""
    p = amd.ConvParams(256, hw, hw, c, 3, 3, c, padding=amd.PADDING_SAME, pad_values=1, dst_type=getattr(amd, dst))
    assert amd.Bconv2dPlan(p).kernel_name() == want


def test_forced_strips_on_another_bank_say_so():
    """Advisor (round 4): stream_strip > 0 on a layer whose filter bank is not the 256-channel one used to end in the unrelated
    'row ring does not fit LDS'."""
    p = amd.ConvParams(2, 8, 64, 128, 3, 3, 128, padding=amd.PADDING_SAME, pad_values=1)
    plan = amd.Bconv2dPlan(p)
    plan.set_option("engine", "stream")
    plan.set_option("stream_strip", "32")
    assert plan.kernel_name() == ""
    assert "256-channel filter bank only" in amd.lib().lce_hip_last_error().decode()


def test_planner_choice_for_run_dual():
    """lce_hip_bconv2d_plan_kernel_name_dual names the kernel run_dual launches: since round 5 the same kernel as run's (the cost
    estimate does not depend on the kind of call; round 4's twin plan is gone)."""
    for hw, c, dst in ((56, 64, amd.I8), (28, 128, amd.I8), (28, 128, amd.F32), (56, 256, amd.F32)):
        p = amd.ConvParams(256, hw, hw, c, 3, 3, c, padding=amd.PADDING_SAME, pad_values=1, dst_type=dst)
        plan = amd.Bconv2dPlan(p)
        assert plan.kernel_name(dual=True) == plan.kernel_name() and plan.kernel_name().startswith("bconv2d_stream<")
    plan.set_option("engine", "mfma")                       # a forced engine holds for both kinds of call
    assert plan.kernel_name(dual=True) == plan.kernel_name() and plan.kernel_name().startswith("bconv2d_mfma")


def test_planner_fallbacks():
    def name(**kw):
        base = dict(batch=256, in_height=28, in_width=28, channels_in=128, filter_height=3, filter_width=3,
                    channels_out=128)
        base.update(kw)
        return amd.Bconv2dPlan(amd.ConvParams(**base)).kernel_name()
    assert name(groups=2) == "bconv2d_mfma_direct<f32,256x64>"         # grouped, 64 channels per group: one group per block
    assert name(groups=4).startswith("bconv2d_tiled<")                 # 32 channels per group: xor-popcount engine
    # a launch of sixteen pixels: round 4 sent it to the xor-popcount engine; any launch the streaming family can run is priced now,
    # and one block of the weight-streaming kernel is the cheapest (batch-1 rows of profiles/r05/engine_sweep_box1.jsonl)
    assert name(batch=1, in_height=4, in_width=4, channels_out=8).startswith("bconv2d_wstream<")
    assert name(batch=1, in_height=4, in_width=4, channels_in=32, channels_out=8).startswith("bconv2d_stream<")   # (32 channels pad to one 64-channel chunk)
    assert name(batch=1, in_height=8, in_width=8, filter_height=5, filter_width=5, channels_out=8).startswith("bconv2d_tiled<")   # 5x5, tiny: still the xor-popcount engine
    assert name(channels_in=2048, batch=8).startswith("bconv2d_mfma<")  # LDS halo too large: workspace GEMM
    # 13x13 outputs: a 128-pixel tile would be 34 % padding -> workspace GEMM (tiles span images)
    assert name(channels_in=100, channels_out=33, stride_height=2, stride_width=2).startswith("bconv2d_mfma<")
    # 27x27 outputs: 6 tiles of 128 pixels pad 5 % -> direct
    assert name(in_height=56, in_width=56, channels_in=100, channels_out=33, stride_height=2,
                stride_width=2).startswith("bconv2d_mfma_direct<")
