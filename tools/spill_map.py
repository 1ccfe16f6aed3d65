#!/usr/bin/env python3
"""Where in a kernel's ISA the scratch (spill) traffic sits: counts of scratch stores / loads bucketed by the number of
MFMAs that precede them.   python tools/spill_map.py <mangled-name-prefix> [-D...]"""
import bisect
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "egoego_release_amd", "csrc")
name = sys.argv[1]
defs = [a for a in sys.argv[2:] if a.startswith("-D")]
os.makedirs("/tmp/_sm", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-save-temps", "-o", "/tmp/_sm/x.so",
                os.path.join(CSRC, "egoego_hip.hip")] + defs, cwd="/tmp/_sm", capture_output=True)
s = open("/tmp/_sm/egoego_hip-hip-amdgcn-amd-amdhsa-gfx950.s").read()
i0 = s.index("\n" + name)
i0 = s.index(":\n", i0)
body = s[i0:s.index(".Lfunc_end", i0)].splitlines()
mf = [i for i, l in enumerate(body) if "v_mfma" in l]
bar = [i for i, l in enumerate(body) if "s_barrier" in l]
def hist(tag):
    b = {}
    for i, l in enumerate(body):
        if tag in l:
            k = bisect.bisect(mf, i)
            b[k] = b.get(k, 0) + 1
    return sorted(b.items())
print(len(body), "lines,", len(mf), "mfma,", len(bar), "barriers")
print("scratch_store by #mfma before:", hist("scratch_store"))
print("scratch_load  by #mfma before:", hist("scratch_load"))
if "--dump" in sys.argv:
    open("/tmp/_sm/kernel.s", "w").write("\n".join(body))
    print("wrote /tmp/_sm/kernel.s")
