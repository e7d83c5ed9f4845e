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
STFT_SRC = os.path.join(PKG_DIR, "csrc", "sot_stft.hip")   # the STFT-magnitude producer: its own translation unit
OSC_SRC = os.path.join(PKG_DIR, "csrc", "sot_osc.hip")     # the oscillator bank
MSS_SRC = os.path.join(PKG_DIR, "csrc", "sot_mss.hip")     # MSSLoss with its gradient in two launches (round 5)
DEPS = [SRC, STFT_SRC, OSC_SRC, MSS_SRC, os.path.join(PKG_DIR, "csrc", "sot_stft_tables.inc"), os.path.join(PKG_DIR, "csrc", "sot_device.hpp"), os.path.join(PKG_DIR, "csrc", "sot_wave_sort.hpp"), os.path.join(PKG_DIR, "csrc", "sot_wave_fft.hpp"), os.path.join(PKG_DIR, "csrc", "sot_forward_full.inc"),
        os.path.join(os.path.dirname(PKG_DIR), "include", "sot_hip.h")]
LIB = os.path.join(PKG_DIR, "libsot_hip.so")

# -ffp-contract=off / -fno-fast-math: the kernels rely on IEEE fp32 division and unfused
# multiply/add to stay bit-compatible with the reference's CPU arithmetic.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
               "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# sot_hip.hip is compiled in parts (-DSOT_PART=<bit>) in parallel and linked into one shared library:
# compile-time-length forward and backward kernels, forward/shared positions (no cutoff, cutoff), forward/per-row positions,
# backward/shared, backward/per-row, everything else, CSR forward, position gradients.
PARTS = (128, 256, 512, 1024, 1, 64, 2, 4, 8, 16, 32, 2048)   # the longest first (bits 7-10: compile-time-geometry kernels)
OBJ_DIR = os.path.join(PKG_DIR, "csrc", "obj")


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


DIGEST = LIB + ".digest"   # sha256 of the sources the library was built from (git-ignored, travels with the .so)


def source_digest() -> str:
    """Content hash of every source the library is compiled from (+ the compiler flags): robust against copies of the
    tree that do not preserve modification times, unlike an mtime comparison."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def is_stale() -> bool:
    """True when libsot_hip.so is missing or was built from other sources than the ones in the tree: a stale library
    behind a changed C signature would be called with mis-marshalled arguments."""
    if not (os.path.exists(LIB) and os.path.exists(DIGEST)):
        return True
    with open(DIGEST) as f:
        return f.read().strip() != source_digest()


def _compile_part(part, extra_flags, verbose: bool, obj_dir: str = None) -> str:
    obj_dir = obj_dir or OBJ_DIR
    if part in ("stft", "osc", "mss"):
        obj = os.path.join(obj_dir, f"sot_{part}.o")
        cmd = [hipcc_path(), *HIPCC_FLAGS, *extra_flags, "-c", "-o", obj, {"stft": STFT_SRC, "osc": OSC_SRC, "mss": MSS_SRC}[part]]
    else:
        obj = os.path.join(obj_dir, f"sot_part{part}.o")
        cmd = [hipcc_path(), *HIPCC_FLAGS, *extra_flags, f"-DSOT_PART={part}", "-c", "-o", obj, SRC]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on part {part}:\n" + res.stdout + res.stderr)
    return obj


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = None) -> str:
    """Compile every kernel instantiation for gfx950 and link libsot_hip.so (parts built in parallel).

    One builder at a time per tree (an exclusive lock on a file next to the library: several ranks that find the library
    stale at once would otherwise compile into the same object files).  The digest is removed before anything is
    overwritten and only written back for a flag-free build into the product path, so a diagnostic build (extra_flags)
    that lands on libsot_hip.so is never mistaken for the product library by is_stale()."""
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    lib = out or LIB
    if not force and out is None and not is_stale():
        return lib
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and out is None and not is_stale():   # another process built it while this one waited
            return lib
        if out is None and os.path.exists(DIGEST):
            os.remove(DIGEST)
        # diagnostic variants (out / extra_flags) get their own object directory: they never mix with the product objects
        obj_dir = OBJ_DIR if (out is None and not extra_flags) else os.path.join(OBJ_DIR, "variant_%d" % os.getpid())
        os.makedirs(obj_dir, exist_ok=True)
        workers = max(1, min(len(PARTS), (os.cpu_count() or 2)))
        with ThreadPoolExecutor(max_workers=workers) as pool:
            objs = list(pool.map(lambda part: _compile_part(part, tuple(extra_flags), verbose, obj_dir), (*PARTS, "stft", "osc", "mss")))
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib + ".tmp", *objs]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
        os.replace(lib + ".tmp", lib)
        if obj_dir != OBJ_DIR:
            shutil.rmtree(obj_dir, ignore_errors=True)
        if out is None and not extra_flags:
            with open(DIGEST, "w") as f:
                f.write(source_digest() + "\n")
    return lib


# ---- the C++ host path of the module (csrc/sot_torch_glue.cpp): a torch extension WITHOUT device code, built in-tree -----------
GLUE_SRC = os.path.join(PKG_DIR, "csrc", "sot_torch_glue.cpp")
GLUE_LIB = os.path.join(PKG_DIR, "_sot_glue.so")
GLUE_DIGEST = GLUE_LIB + ".digest"


def glue_digest() -> str:
    import hashlib
    import torch
    h = hashlib.sha256()
    for d in (GLUE_SRC, os.path.join(os.path.dirname(PKG_DIR), "include", "sot_hip.h")):
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    h.update(torch.__version__.encode())
    return h.hexdigest()


def glue_is_stale() -> bool:
    if not (os.path.exists(GLUE_LIB) and os.path.exists(GLUE_DIGEST)):
        return True
    with open(GLUE_DIGEST) as f:
        return f.read().strip() != glue_digest()


def build_glue(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/sot_torch_glue.cpp against this interpreter's torch (host compiler only; c10_hip supplies the current HIP
    stream) into _sot_glue.so next to this file.  Same locking / digest discipline as build()."""
    import fcntl
    import torch
    from torch.utils import cpp_extension
    if not force and not glue_is_stale():
        return GLUE_LIB
    work = os.path.join(OBJ_DIR, "glue")
    os.makedirs(work, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".glue.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not glue_is_stale():
            return GLUE_LIB
        if os.path.exists(GLUE_DIGEST):
            os.remove(GLUE_DIGEST)
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
        cpp_extension.load(name="_sot_glue", sources=[GLUE_SRC], build_directory=work, verbose=verbose, is_python_module=False,
                           extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-Wno-unused-function"],
                           extra_include_paths=[os.path.join(rocm, "include")],
                           extra_ldflags=["-ldl", f"-L{torch_lib}", "-lc10_hip", "-ltorch_hip", f"-Wl,-rpath,{torch_lib}"], with_cuda=False)
        shutil.copyfile(os.path.join(work, "_sot_glue.so"), GLUE_LIB + ".tmp")
        os.replace(GLUE_LIB + ".tmp", GLUE_LIB)
        with open(GLUE_DIGEST, "w") as f:
            f.write(glue_digest() + "\n")
    return GLUE_LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_glue(force=True, verbose=True))
