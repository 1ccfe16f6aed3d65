#!/usr/bin/env python3
"""attn8_kernel (csrc/attention.h) loads its Q fragments with inline-assembly `global_load_dwordx4`, which hipcc does not track: the kernel
waits for them by hand (`s_waitcnt vmcnt(N)`) before it lets the compiler read them.  This check compiles the library to gfx950 assembly and
verifies that NO instruction reads or writes a destination register of such a load between the load and the next `s_waitcnt vmcnt` — i.e.
that register allocation put no copy in between (a copy there would move stale data).  ~70 s of hipcc, no GPU.

    python tools/check_untracked_loads.py        (exit code 1 on a violation)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "egoego_release_amd", "csrc")
KERNELS = ["_Z12attn8_kernelILi7ELi2EEv8AttnArgs"]


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main():
    out = "/tmp/_egoego_untracked.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-S", "--cuda-device-only", "-o", out,
                    os.path.join(CSRC, "egoego_hip.hip")], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    s = open(out).read()
    bad = 0
    for name in KERNELS:
        i = s.index("\n" + name + ":")
        lines = [ln.strip().split(";")[0].strip() for ln in s[i:s.index("s_endpgm", i)].splitlines()]
        lines = [ln for ln in lines if ln and not ln.startswith(".")]
        n_loads = 0
        for n, ln in enumerate(lines):
            if not ln.startswith("global_load_dwordx4") or "lds" in ln:
                continue
            n_loads += 1
            dest = regs(ln.split()[1].rstrip(","))
            for m in lines[n + 1:]:
                if m.startswith("s_waitcnt") and "vmcnt" in m:
                    break
                used = set()
                for o in re.findall(r"v\[\d+:\d+\]|v\d+", m):
                    used |= regs(o)
                if used & dest:
                    bad += 1
                    print(f"{name}: `{m}` touches the destination of `{ln}` before a vmcnt wait")
                    break
        print(f"{name}: {n_loads} untracked loads checked")
        if n_loads == 0:
            bad += 1
            print("  (expected some: the kernel changed?)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
