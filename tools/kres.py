#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of one HIP source, from the compiler's kernel-resource-usage remarks.
Usage: python tools/kres.py faceposegenerator_amd/csrc/idb_gemm.hip [substring filter]"""
import os
import re
import subprocess
import sys

src = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + inc, "--cuda-device-only",
                    "-Rpass-analysis=kernel-resource-usage", "-S", src, "-o", "/tmp/kres.s"], capture_output=True, text=True)
t = r.stderr
for l in t.split("\n"):
    if "error" in l:
        print(l)
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].split(" ")[0]
    if flt not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    print(name[:72].ljust(72), "V", g("VGPRs"), "A", g("AGPRs"), "S", g("SGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"),
          "occ", g(r"Occupancy \[waves/SIMD\]"), "lds", g(r"LDS Size \[bytes/block\]"))
