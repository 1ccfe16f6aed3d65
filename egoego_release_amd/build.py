"""Build libegoego_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libegoego_hip.so")
PERFDEBUG_DIR = os.path.join(os.path.dirname(PKG), "tools", "_build")
SOURCES = ["egoego_hip.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "egoego_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, perfdebug=False, defines=(), tag="", variant=False):
    """Compile csrc/*.hip -> egoego_release_amd/libegoego_hip.so (gfx950 only).

    perfdebug=True builds tools/_build/libegoego_hip_perfdebug.so instead: the same sources with -DEGOEGO_PERFDEBUG (per-block
    timestamps, stage ablation; plus any extra `defines` such as EGOEGO_ABLATE_MAINLOOP=1) for tools/*_trace.py.
    The product library contains none of that."""
    out = LIB
    if perfdebug:  # tools-only builds stay out of the package directory
        os.makedirs(PERFDEBUG_DIR, exist_ok=True)
        out = os.path.join(PERFDEBUG_DIR, "libegoego_hip_perfdebug" + (f"_{tag}" if tag else "") + ".so")
    if not force and not perfdebug and not is_stale():
        return out
    # -ffp-contract=on: a multiply and an add fuse only where ONE source expression holds both (hipcc's default, "fast", also
    # fuses across statements after inlining — then the bits of an epilogue depend on the kernel it was inlined into, and
    # the same row computed by two tilings, e.g. a 32-window shard and the full batch, may differ in the last place)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-shared", "-fPIC", "-o", out]
    if perfdebug:
        # (--variant: the PRODUCT flags plus the extra defines — an A/B build of one knob, without the trace code)
        cmd += ([] if variant else ["-DEGOEGO_PERFDEBUG"]) + [f"-D{d}" for d in defines]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, perfdebug="--perfdebug" in sys.argv or "--variant" in sys.argv, variant="--variant" in sys.argv,
                defines=[a[2:] for a in sys.argv if a.startswith("-D")],
                tag=next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--tag=")), "")))
