#!/usr/bin/env python3
"""The 20-state whole-list kernel (partials_aa_fused.hip) keeps its slots in the accumulation registers
a0..a109, which only its own inline assembly may touch.  This checks that no instruction outside an
inline-assembly block names one of them (nor M0, which the kernels' LDS-DMA assembly owns without saving it) -- in the assembly file given as argument (the Makefile passes the
-save-temps output of the compilation that makes the object: a hit fails the build), or, without one, in a
compilation of its own with the Makefile's flags.  Scratch use by the kernel fails too: what is spilled at 128
registers is what the compiler next parks in the accumulation registers.
Exit status 0 = clean.  (hipcc cross-compiles: no GPU needed.)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    # with a path: the assembly the build has just produced (Makefile, -save-temps of the very compilation that
    # makes the object); without: compile here with the Makefile's flags for gfx950
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    allow_scratch = "--allow-scratch" in sys.argv   # (tool builds with cycle stamps keep their tables in scratch)
    report_only = "--report-only" in sys.argv       # (tool builds: their printf at the end of a tile uses the file freely)
    if args:
        text = open(args[0]).read()
    else:
        src = os.path.join(ROOT, "libpll_amd", "csrc", "hip", "partials_aa_fused.hip")
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "af.s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
                   "-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0",
                   "-I" + os.path.join(ROOT, "include"),
                   "-I" + os.path.join(ROOT, "libpll_amd", "csrc", "hip"), "-S", "--cuda-device-only", "-o", out, src]
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            text = open(out).read()
    bad, in_asm, kernel = [], False, None
    bad_m0 = []   # M0 belongs to the kernels' LDS-DMA assembly (af_dma_run): nothing else may read or write it
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
            if re.search(r"\bm0\b", code):
                bad_m0.append((n, line.strip()))
        if line.strip() == "s_endpgm":
            kernel = None
    scratch = [l.strip() for l in text.splitlines() if ".private_segment_fixed_size:" in l and l.split(":")[1].strip() != "0"]
    # (k_af_prepare and k_aa_fused<...> are all the kernels of the file: none may use scratch)
    print("%d kernels checked, %d instructions outside the slot assembly touch a0..a109, %d touch m0, %d kernels with scratch"
          % (kernels, len(bad), len(bad_m0), len(scratch)))
    for n, line in (bad + bad_m0)[:10]:
        print("  line %d: %s" % (n, line))
    if report_only:
        return 0
    return 1 if bad or bad_m0 or (scratch and not allow_scratch) or kernels == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
