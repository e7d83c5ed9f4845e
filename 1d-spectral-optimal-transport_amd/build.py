"""Builds libsot_hip.so (the C-ABI HIP library of include/sot_hip.h) for gfx950 with hipcc.

In-tree build: the .so lands next to this file (git-ignored, but it travels to the GPU box with the
gpurun snapshot).  hipcc cross-compiles without a GPU, so this also runs in the CPU-only container.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG_DIR, "csrc", "sot_hip.hip")
DEPS = [SRC, os.path.join(PKG_DIR, "csrc", "sot_device.hpp"),
        os.path.join(os.path.dirname(PKG_DIR), "include", "sot_hip.h")]
LIB = os.path.join(PKG_DIR, "libsot_hip.so")

# -ffp-contract=off / -fno-fast-math: the kernels rely on IEEE fp32 division and unfused
# multiply/add to stay bit-compatible with the reference's CPU arithmetic.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-fno-fast-math", "-Wall"]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), *HIPCC_FLAGS, *extra_flags, "-o", LIB + ".tmp", SRC]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
