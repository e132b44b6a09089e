"""The fence around the measuring aids in the product sources (csrc/lce_experiments.h): every LCE_* name the
preprocessor tests anywhere in the product tree is either registered there -- and then a compile error without
-DLCE_EXPERIMENT, which the product Makefile never passes -- or one of the structural names."""
import glob
import os
import re
import subprocess

import pytest

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


@pytest.mark.parametrize("family,least", [("lce_tu_stream_*", 90), ("lce_tu_wstream_*", 84), ("lce_tu_pointwise", 40)])
def test_no_streaming_kernel_instance_uses_scratch_memory(family, least):
    """Round 5 (review item 7): the code objects of the built product library -- their own metadata, tools/co_resources.py -- hold no
    kernel of the streaming families with a private-segment (scratch) size above 0: no VGPR spill goes to memory.  (Round 6: the two
    streaming families' tables are cut into several translation units -- all of a family's objects are read.)"""
    import subprocess
    import sys
    objs = sorted(glob.glob(os.path.join(ROOT, "compute-engine_amd", "csrc", "obj", family + ".o")))
    if not objs or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("the product library's objects / the LLVM tools are not here")
    rows = []
    for obj in objs:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "co_resources.py"), obj], capture_output=True, text=True, check=True).stdout
        rows += [l for l in out.splitlines() if " scratch " in l]
    assert len(rows) >= least, (len(rows), rows[-3:])
    bad = [l for l in rows if int(l.split(" scratch ")[1].split()[0]) != 0]
    assert not bad, "\n".join(bad)


def test_every_kernel_is_emitted_once():
    """Round 6: until then a non-template inline function in each dispatch header named ALL parts of a family, so every translation
    unit that included the header instantiated -- and emitted -- every kernel of the family (the block GEMM was in the library three
    times, 19.6 MB).  Each kernel symbol now appears in exactly one object, and the library stays under 15 MB."""
    import subprocess
    objs = sorted(glob.glob(os.path.join(ROOT, "compute-engine_amd", "csrc", "obj", "lce_tu_*.o")))
    import shutil
    nm = shutil.which("nm")
    if not objs or not nm:
        pytest.skip("the product library's objects / the LLVM tools are not here")
    seen = {}
    for obj in objs:
        for line in subprocess.run([nm, "--defined-only", obj], capture_output=True, text=True, check=True).stdout.splitlines():
            parts = line.split()
            # host-side launch stubs of the kernels: __device_stub__<mangled kernel>
            if len(parts) == 3 and "__device_stub__" in parts[2] and "bconv2d" in parts[2]:
                seen.setdefault(parts[2], []).append(os.path.basename(obj))
    assert len(seen) >= 350, len(seen)
    twice = {k: v for k, v in seen.items() if len(v) > 1}
    assert not twice, list(twice.items())[:3]
    lib = os.path.join(ROOT, "compute-engine_amd", "csrc", "liblce_hip.so")
    assert os.path.getsize(lib) < 15 * 1024 * 1024, os.path.getsize(lib)
