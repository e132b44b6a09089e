// Registry of every build-time switch the kernels, the planner and the C ABI know, and the fence around them.
//
// The PRODUCT library (csrc/Makefile, -DLCE_PRODUCT_BUILD) is built with NONE of them.  All of them are measuring aids:
// timing ablations that remove a piece of a kernel (the results are then WRONG by construction -- only the clock is
// read), A/B variants of a code path, per-block time stamps for the phase tools.  One stray -D would otherwise ship a
// library that computes garbage, so:
//   * a switch from this list without -DLCE_EXPERIMENT is a compile error;
//   * -DLCE_EXPERIMENT together with -DLCE_PRODUCT_BUILD is a compile error;
//   * lce_hip_build_flavor() (include/lce_hip.h) reports which of the two a loaded library is, and
//     tests/test_cabi_host.py asserts that the in-tree one says "product";
//   * tests/test_build_hygiene.py greps the sources: every LCE_* name tested by the preprocessor must be listed here
//     (or be the structural name below), so a new switch cannot bypass the fence.
// tools/build_exp.sh builds experiment variants (it passes -DLCE_EXPERIMENT itself) into build_exp/, never in-tree.
//
// Structural, NOT experiments: LCE_USE_SYSTEM_TFLITE (csrc/tflite: real TensorFlow Lite headers instead of the layout
// mirror).
#pragma once

// ---- "results are wrong" timing ablations ----
#if defined(LCE_ABL_NODMA) || defined(LCE_ABL_NOBAR) || defined(LCE_ABL_NOFRAG) || defined(LCE_ABL_NOSTORE) || \
    defined(LCE_ABL_NOSLEEP) || defined(LCE_ST_NOFRAG) || defined(LCE_ST_NOEPI) || defined(LCE_ST_NOEPI_A) ||  \
    defined(LCE_ST_NOEPI_B) || defined(LCE_ST_NOEPI_C) || defined(LCE_ST_NOPROD) || defined(LCE_PW_NOSTORE) ||   \
    defined(LCE_ST_NOBANK)
#define LCE_HAS_EXPERIMENT_SWITCH 1
#endif
// ---- time-stamp builds for the phase / timeline tools (results right, kernels slower, extra debug exports), and the
//      single-translation-unit build they need (tools/build_exp.sh; the product is one translation unit per family) ----
#if defined(LCE_TIMELINE) || defined(LCE_PHASES) || defined(LCE_STREAM_PHASES) || defined(LCE_PW_PHASES) || defined(LCE_UNITY)
#define LCE_HAS_EXPERIMENT_SWITCH 1
#endif
// ---- A/B variants of a code path (results right) ----
#if defined(LCE_NO_PK_F32) || defined(LCE_MFMA_SCALED) || defined(LCE_STREAM_LDS_WAIT) || defined(LCE_STREAM_PK_F32) || \
    defined(LCE_NO_HALO_SKIP) || defined(LCE_NO_HALO_FULLWORDS) || defined(LCE_STREAM_NO_SKEW) || defined(LCE_WORDS_THROUGH) || \
    defined(LCE_STREAM_AUTO_STRIDED) || defined(LCE_STREAM_AUTO_LOWK) || defined(LCE_COST_TUNABLE)
#define LCE_HAS_EXPERIMENT_SWITCH 1
#endif
// ---- tunables with a built-in default (the sources #define them when the command line does not) ----
#if defined(LCE_STORE_AUX) || defined(LCE_STORE_THROUGH_AUX) || defined(LCE_STORE8_AUX) || defined(LCE_STOREW_AUX) || \
    defined(LCE_STORE_PACE) || defined(LCE_STREAM_MIN_STEPS)
#define LCE_HAS_EXPERIMENT_SWITCH 1
#endif

#if defined(LCE_HAS_EXPERIMENT_SWITCH) && !defined(LCE_EXPERIMENT)
#error "an experiment / ablation switch is defined without -DLCE_EXPERIMENT: the product library takes none (see lce_experiments.h; tools/build_exp.sh builds variants)"
#endif
#if defined(LCE_EXPERIMENT) && defined(LCE_PRODUCT_BUILD)
#error "csrc/Makefile builds the product library: -DLCE_EXPERIMENT does not belong on its command line"
#endif

#ifdef LCE_EXPERIMENT
#define LCE_BUILD_FLAVOR "experiment"
#else
#define LCE_BUILD_FLAVOR "product"
#endif
