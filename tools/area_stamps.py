"""Diagnostic (build with `python tools/build_variants.py stamps:-DSOT_STAMPS`): where thread 0 of one workgroup of the merge-free
p = 1 kernel spends its cycles, phase by phase, summed over its rows in a full-size launch (shares; the stamp fences cost a little)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "tools", "ablate_libs", "stamps.so")
os.environ["SOT_LIB_PATH"] = lib
import torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device("cuda:0"); B, N = 8192, 2048
g = torch.Generator(device=dev).manual_seed(0)
sets = [(torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)) for _ in range(6)]
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
plan = nat.PositionPlan(pos, pos2)
for i in range(300):
    nat.forward_rows(*sets[i % 6], pos, pos2, 1.0, 8, plan)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 64)()
ctypes.CDLL(lib).sot_debug_read_stamps(out, 64)
names = {1: "wait for the row's loads + staging stores", 2: "barrier 1", 3: "chunk sums", 4: "barrier 2", 5: "fold (waves 0,1) / owner reads",
         6: "barrier 3", 7: "division + fp64 accumulation + wave scans", 8: "barrier 4", 9: "wave totals, CDF values, next loads issued",
         10: "area + wave sum + barrier 5 + store"}
rows = out[0]
tot = sum(out[i] for i in names)
print(f"{rows} rows of one workgroup: {tot / rows:.0f} cycles per row")
for i, n in names.items():
    print(f"  {n:48s} {out[i] / rows:8.0f} cycles  {100 * out[i] / tot:5.1f} %")
