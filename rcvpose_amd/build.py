"""Builds librcvpose_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m rcvpose_amd.build [--force]

The library is compiled from rcvpose_amd/csrc/*.hip; the C ABI it exports is include/rcvpose_hip.h.
hipcc cross-compiles for gfx950 without a GPU, so this also runs in the CPU-only build container.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librcvpose_hip.so")
ARCH = "gfx950"

# accumulator.hip must round like numpy (no FMA contraction); the conv kernels accumulate in MFMA.
SOURCES = {
    "core.hip": [],
    "accumulator.hip": ["-ffp-contract=off"],
    "fcn.hip": [],
    "fcn_f16.hip": [],   # fcn.hip again with IEEE half as the 16-bit storage type
    "fcn_x3.hip": [],    # ... and as the split-precision engine (half hi / lo pairs, three MFMAs per product)
    "pose.hip": ["-ffp-contract=off"],
    "icp.hip": ["-ffp-contract=off"],
}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
          "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=False):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "rcvpose_hip.h"))
    objs = []
    obj_dir = os.path.join(HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    for src, extra in SOURCES.items():
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [sp] + headers + ([os.path.join(CSRC, "fcn.hip")] if src in ("fcn_f16.hip", "fcn_x3.hip") else [])):
            cmd = [_hipcc()] + COMMON + extra + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    if force or procs or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
