#!/usr/bin/env python
"""Per-kernel VGPR / AGPR / LDS / spill figures of the gfx950 code objects in csrc/*.o (no GPU needed).

usage: tools/kernel_resources.py [substring ...]   -- only kernels whose demangled name contains every substring
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    import glob
    csrc = os.path.join(ROOT, "alphazero.jl_amd", "csrc")
    notes = ""
    with tempfile.TemporaryDirectory() as d:
        for obj in sorted(glob.glob(os.path.join(csrc, "*.o"))):      # one fat binary per translation unit
            fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "az.co")
            subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
            r = subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                "--input=" + fat, "--output=" + co, "--unbundle"], capture_output=True)
            if r.returncode != 0:
                continue                                                # a unit without kernels (net.o)
            notes += subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    rows = []
    for e in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", e).group(1)
        rows.append((g("name"), g("vgpr_count"), re.match(r":?\s*(\d+)", e).group(1), g("group_segment_fixed_size"),
                     g("vgpr_spill_count"), g("private_segment_fixed_size"), g("sgpr_count")))
    dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
    for r, name in zip(rows, dem):
        name = name.replace("void ", "")
        if all(s in name for s in sys.argv[1:]):
            print("%-100s vgpr %3s agpr %3s sgpr %3s lds %6s spill %s scratch %s" % (name[:100], r[1], r[2], r[6], r[3], r[4], r[5]))


if __name__ == "__main__":
    main()
