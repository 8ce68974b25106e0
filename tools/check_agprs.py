#!/usr/bin/env python3
"""The 20-state whole-list kernel (partials_aa_fused.hip) keeps its slots in the accumulation registers
a0..a109, which only its own inline assembly may touch.  This compiles the file to assembly with the
Makefile's flags and checks that no instruction outside an inline-assembly block names one of them.
Exit status 0 = clean.  (hipcc cross-compiles: no GPU needed.)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.join(ROOT, "libpll_amd", "csrc", "hip", "partials_aa_fused.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "af.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
               "-mllvm", "-amdgpu-mfma-vgpr-form", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "libpll_amd", "csrc", "hip"), "-S", "--cuda-device-only", "-o", out, src]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    bad, in_asm, kernel = [], False, None
    reg = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
    kernels = 0
    for n, line in enumerate(text.splitlines(), 1):
        if "k_aa_fused" in line and not line.startswith(("\t", " ", ".", ";")) and ":" in line:
            kernel = line.strip()
            kernels += 1
        if ";#ASMSTART" in line:
            in_asm = True
        elif ";#ASMEND" in line:
            in_asm = False
        elif kernel and not in_asm and not line.lstrip().startswith((";", ".")):
            code = line.split(";")[0]
            for m in reg.finditer(code):
                lo = int(m.group(1) or m.group(2))
                if lo < 110:
                    bad.append((n, line.strip()))
        if line.strip() == "s_endpgm":
            kernel = None
    print("%d kernels checked, %d instructions outside the slot assembly touch a0..a109" % (kernels, len(bad)))
    for n, line in bad[:10]:
        print("  line %d: %s" % (n, line))
    return 1 if bad or kernels == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
