#!/usr/bin/env python3
"""Register / spill / LDS usage of every kernel in libegoego_hip (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py [filter-substring] [-D...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "egoego_release_amd", "csrc")
flt = [a for a in sys.argv[1:] if not a.startswith("-D")]
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-shared", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
       "-o", "/tmp/_kr.so", os.path.join(CSRC, "egoego_hip.hip")] + defs
out = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|TotalSGPRs|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\S+)", ln)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:110]
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
print(f"{'VGPR':>5} {'AGPR':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'occ':>3}  kernel")
for name, r in rows.items():
    if flt and not any(f in name for f in flt):
        continue
    print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('VGPRs Spill', '?'):>6} {r.get('SGPRs Spill', '?'):>6} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>7} {r.get('Occupancy [waves/SIMD]', '?'):>3}  {name}")
