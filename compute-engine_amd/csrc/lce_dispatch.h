// Kernel-instance lookup, all families at once: what the host simulation used by the CPU-only tests (tests/hostsim/)
// and the single-translation-unit experiment builds (tools/build_exp.sh, -DLCE_UNITY) include.  The product build
// compiles one family per translation unit (lce_tu_*.hip, lce_kernel_types.h).
#pragma once
#include "lce_dispatch_valu.h"
#include "lce_dispatch_mfma.h"
#include "lce_dispatch_pointwise.h"
#include "lce_dispatch_stream.h"
#include "lce_dispatch_wstream.h"
