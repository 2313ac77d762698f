#!/usr/bin/env python
"""Developer tool: LDS bytes, scratch bytes and VGPRs of every kernel, from `hipcc -S --cuda-device-only` of csrc/*.hip.

Why it exists: round 3 found k_syrkd_sliced<float, 128, MODE=2> with 163 840 bytes of LDS for a 131 072-byte tile -- a
16-byte record kept in an alloca that the compiler promoted to LDS (32 bytes per thread) because a select of loads had been
folded into a load at a selected address.  Nothing warns about that; the numbers below show it (LDS larger than the
kernel's own arrays, or scratch > 0).  Usage: python tools/kernel_resources.py [spmm spgemm gram bsr dense handle]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SRC = os.path.join(ROOT, "sparse_dot_amd", "csrc")


def main():
    files = sys.argv[1:] or ["spmm", "spgemm", "gram", "bsr", "dense", "handle"]
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for f in files:
            out = os.path.join(tmp, f + ".s")
            procs.append((f, out, subprocess.Popen(
                ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I" + SRC,
                 "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-x", "hip", os.path.join(SRC, f + ".hip"),
                 "-o", out], stderr=subprocess.DEVNULL)))
        print("%-8s %8s %8s %5s  %s" % ("file", "LDS", "scratch", "VGPR", "kernel"))
        for f, out, p in procs:
            if p.wait() != 0:
                print("%-8s compile failed" % f)
                continue
            s = open(out).read()
            for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", s, re.S):
                body = m.group(2)
                lds = int(re.search(r"group_segment_fixed_size (\d+)", body).group(1))
                scratch = int(re.search(r"private_segment_fixed_size (\d+)", body).group(1))
                vgpr = int(re.search(r"next_free_vgpr (\d+)", body).group(1))
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(.*", "", name)
                flag = "  <-- scratch" if scratch else ""
                print("%-8s %8d %8d %5d  %s%s" % (f, lds, scratch, vgpr, name[:110], flag))


if __name__ == "__main__":
    main()
