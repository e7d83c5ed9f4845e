"""Diagnostic (build with `python tools/build_variants.py stamps:-DSOT_STAMPS`): where thread 0 of one workgroup of the training-form
kernel (loss + gradient w.r.t. the estimate, paper mode) spends its cycles, phase by phase, summed over its rows in a full-size launch.
Usage: python tools/train_stamps.py [B] [N]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "tools", "ablate_libs", os.environ.get("STAMPS_LIB", "stamps") + ".so")
os.environ["SOT_LIB_PATH"] = lib
import torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device("cuda:0")
B, N = (int(v) for v in (sys.argv[1:3] + [8192, 2048][len(sys.argv) - 1:]))
g = torch.Generator(device=dev).manual_seed(0)
sets = [(torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)) for _ in range(6)]
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
plan = nat.PositionPlan(pos, pos2)
for i in range(300):
    nat.loss_and_grad(*sets[i % 6], pos, pos2, 2.0, 15, plan)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 64)()
ctypes.CDLL(lib).sot_debug_read_stamps(out, 64)
names = {1: "wait for the row's loads + staging stores", 2: "barrier 1", 3: "chunk sums", 4: "barrier 2", 5: "fold (waves 0,1) / owner reads",
         6: "barrier 3", 7: "division + fp64 accumulation + wave scans", 8: "barrier 4", 9: "wave totals, CDF values to LDS, next loads, barrier",
         10: "partition search + gradient walk", 11: "loss wave sum + barrier", 12: "gradient reads, fp64 reverse sums, scans, exchange",
         13: "dot products, output arithmetic, stores", 14: "last barrier"}
rows = out[0]
tot = sum(out[i] for i in names)
print(f"{rows} rows of one workgroup: {tot / rows:.0f} cycles per row")
for i, n in names.items():
    print(f"  {n:56s} {out[i] / rows:8.0f} cycles  {100 * out[i] / tot:5.1f} %")
