#!/usr/bin/env python
"""Per-kernel ISA comparison of two builds of the product library's translation units (no GPU needed).
usage: isa_diff.py <obj dir A> <obj dir B> [tu ...]
For every kernel of every translation unit: identical instruction stream or not (addresses and encodings ignored), and
for the different ones the instruction counts and the number of v_accvgpr_* copies on either side."""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TUS = ["lce_tu_valu", "lce_tu_mfma_ws", "lce_tu_mfma_direct", "lce_tu_mfma_2d", "lce_tu_pointwise"] + ["lce_tu_stream_" + p for p in ("f32", "f32_clamp", "i8", "i8_floor", "bitpacked")]


def disassemble(obj, tmp):
    fb, co = os.path.join(tmp, "x.fb"), os.path.join(tmp, "x.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", obj, fb + ".copy.o"], check=True)   # (an output file: objcopy would rewrite its input otherwise)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    out, cur = collections.OrderedDict(), None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur and line.strip():
            out[cur].append(line.split("//")[0].strip())
    return out


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


def main():
    a_dir, b_dir = sys.argv[1:3]
    tus = sys.argv[3:] or TUS
    with tempfile.TemporaryDirectory() as tmp:
        for tu in tus:
            a, b = disassemble(os.path.join(a_dir, tu + ".o"), tmp), disassemble(os.path.join(b_dir, tu + ".o"), tmp)
            common = [k for k in a if k in b]
            diff = [k for k in common if a[k] != b[k]]
            print(f"{tu}: {len(common)} kernels in both builds, {len(common) - len(diff)} identical, {len(diff)} different"
                  + (f"; only in A: {len(a) - len(common)}, only in B: {len(b) - len(common)}" if len(a) != len(b) else ""))
            for k in diff:
                print("    %s\n        instructions %d vs %d, v_accvgpr copies %d vs %d" % (
                    demangle(k)[:150], len(a[k]), len(b[k]), sum("v_accvgpr" in l for l in a[k]), sum("v_accvgpr" in l for l in b[k])))


if __name__ == "__main__":
    main()
