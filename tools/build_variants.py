"""Builds diagnostic variants of the library into tools/ablate_libs/<name>.so (full-row kernels + C ABI only):
    python tools/build_variants.py name1:-DFOO=1,-DBAR name2: ...
Run them on the GPU box with tools/ab_probe.py name1 name2 ... (interleaved A/B timing).  VARIANT_PART: the SOT_PART bit mask of kernel
families to compile (default 144 = full-row kernels; add 32 for the CSR forward)."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sot_amd
LIBDIR = os.path.join(ROOT, "tools", "ablate_libs")
os.makedirs(LIBDIR, exist_ok=True)


def one(spec):
    name, _, flags = spec.partition(":")
    out = os.path.join(LIBDIR, name + ".so")
    obj = os.path.join(LIBDIR, name + ".o")
    hipcc = sot_amd.build.hipcc_path()
    r = subprocess.run([hipcc, *sot_amd.build.HIPCC_FLAGS, "-DSOT_PART=" + os.environ.get("VARIANT_PART", "144"), "-DSOT_STUB_MISSING_PARTS", *[f for f in flags.split(",") if f],
                        "-c", "-o", obj, sot_amd.build.SRC], capture_output=True, text=True)
    if r.returncode == 0:   # + the product's STFT / oscillator objects (the binding resolves every exported symbol)
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj, os.path.join(sot_amd.build.OBJ_DIR, "sot_stft.o"),
                            os.path.join(sot_amd.build.OBJ_DIR, "sot_osc.o")], capture_output=True, text=True)
    return name, r.returncode, r.stderr[-2000:]


with ThreadPoolExecutor(4) as ex:
    for name, rc, err in ex.map(one, sys.argv[1:]):
        print(name, "ok" if rc == 0 else "FAILED\n" + err)
