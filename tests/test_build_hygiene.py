"""The fence around the measuring aids in the product sources (csrc/lce_experiments.h): every LCE_* name the
preprocessor tests anywhere in the product tree is either registered there -- and then a compile error without
-DLCE_EXPERIMENT, which the product Makefile never passes -- or one of the structural names."""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "compute-engine_amd", "csrc")
STRUCTURAL = {"LCE_USE_SYSTEM_TFLITE", "LCE_EXPERIMENT", "LCE_PRODUCT_BUILD", "LCE_HAS_EXPERIMENT_SWITCH"}


def _sources():
    out = []
    for pat in ("*.h", "*.hip", "*.cpp", "*.cc", "tflite/*.h", "tflite/*.cc"):
        out += glob.glob(os.path.join(CSRC, pat))
    out += glob.glob(os.path.join(ROOT, "include", "*.h"))
    return [f for f in out if os.path.basename(f) != "lce_experiments.h"]


def _tested_names(text):
    names = set()
    for line in text.splitlines():
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif)\b(.*)", line)
        if m:
            names |= set(re.findall(r"\bLCE_[A-Z0-9_]+\b", m.group(2)))
    return names


def test_every_switch_in_the_product_sources_is_registered():
    registry = open(os.path.join(CSRC, "lce_experiments.h")).read()
    registered = set(re.findall(r"defined\((LCE_[A-Z0-9_]+)\)", registry))
    used = set()
    for f in _sources():
        used |= _tested_names(open(f).read())
    used -= {n for n in used if n.endswith("_H_") or n.endswith("_H")}      # include guards
    unregistered = used - registered - STRUCTURAL
    assert not unregistered, "switches that bypass csrc/lce_experiments.h: %s" % sorted(unregistered)
    assert len(registered) >= 25


def test_the_product_makefile_takes_no_switch_and_says_so():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "-DLCE_PRODUCT_BUILD" in mk
    assert not re.findall(r"-DLCE_(?!PRODUCT_BUILD)[A-Z0-9_]+", mk)
    exp = open(os.path.join(ROOT, "tools", "build_exp.sh")).read()
    assert "-DLCE_EXPERIMENT" in exp


def test_a_stray_switch_is_a_compile_error():
    """Host-side compile of the registry alone: any registered switch without -DLCE_EXPERIMENT stops the build, and so
    does -DLCE_EXPERIMENT in a product build."""
    hdr = os.path.join(CSRC, "lce_experiments.h")
    def cc(*defs):
        return subprocess.run(["g++", "-fsyntax-only", "-x", "c++", *defs, hdr], capture_output=True, text=True)
    assert cc("-DLCE_PRODUCT_BUILD").returncode == 0
    for sw in ("LCE_ST_NOEPI", "LCE_ABL_NODMA", "LCE_PW_NOSTORE", "LCE_STORE_AUX=16", "LCE_MFMA_SCALED", "LCE_PHASES"):
        r = cc("-DLCE_PRODUCT_BUILD", "-D" + sw)
        assert r.returncode != 0 and "LCE_EXPERIMENT" in r.stderr, sw
        assert cc("-DLCE_EXPERIMENT", "-D" + sw).returncode == 0, sw
    assert cc("-DLCE_PRODUCT_BUILD", "-DLCE_EXPERIMENT").returncode != 0
