"""Diagnostic: builds libsot_hip_stamps.so (-DSOT_STAMPS) and prints where one row of workgroup 0 spends
its cycles (shares, not absolute speed: the stamp fences forbid overlaps the real kernel has)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sot_amd  # noqa: E402
from sot_amd import _native as nat  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "p1"
B, N = 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 2048
lib = os.path.join(ROOT, "gpurun_out", "libsot_hip_stamps.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.run([sot_amd.build.hipcc_path(), *sot_amd.build.HIPCC_FLAGS, "-shared", "-DSOT_STAMPS", "-DSOT_PART=17", "-DSOT_STUB_MISSING_PARTS", "-o", lib,
                sot_amd.build.SRC], check=True)  # whole-file diagnostic build: shared-position forward + misc only
sot_amd.build.LIB = lib
nat._lib = None
h = nat.load(build_if_missing=False)
from sot_amd.losses import Wasserstein1D  # noqa: E402

dev = torch.device("cuda:0")
MODES = {"p1": dict(p=1), "cutoff": dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)}
mod = Wasserstein1D(**MODES[mode]).to(dev)
x, y = torch.rand(B, N, device=dev), torch.rand(B, N, device=dev)
pos = torch.linspace(0, 1, N, device=dev)
for _ in range(5):
    mod(x, y, x_pos=pos, y_pos=pos)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 64)()
raw = ctypes.CDLL(lib)
raw.sot_debug_read_stamps(out, 64)
names = ["store+prefetch+B1", "chunk sums+B2", "columns+B3", "fold+div+local scan+wave scan", "B4+offsets+write CDF+B5",
         "merge_path", "merge walk", "reduce+B6+store"]
tot = out[8] - out[0]
print(f"mode={mode} N={N}: one row of WG0 = {tot} cycles")
for i, nm in enumerate(names):
    d = out[i + 1] - out[i]
    print(f"  {nm:34s} {d:8d}  {100.0 * d / tot:5.1f}%")

for name, base in (("first WG", 32), ("last WG", 48)):
    w = [out[base + i] for i in range(14)]
    rows = [w[i] for i in range(2, 14) if w[i]]
    print(f"{name}: setup {w[1] - w[0]} cycles; rows: " + " ".join(str(b - a) for a, b in zip([w[1]] + rows[:-1], rows)) + f"; total {rows[-1] - w[0]}")
print("first WG entry -> last WG exit:", max(out[48 + i] for i in range(2, 14)) - out[32])
