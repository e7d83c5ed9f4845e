"""Phase stamps of the compile-time-length kernels in a full-size launch (build: python tools/build_variants.py stamps:-DSOT_STAMPS).
Thread 0 of ONE workgroup (the middle one of the grid) adds up the shader clocks between consecutive checkpoints over all of its rows;
printed per row.  Shares, not absolute speed: the stamp fences forbid overlaps the real kernel has (cdna_hip_programming.md, In-kernel stamps).
Usage: python tools/r4/phase_stamps.py <fwd|area|lg> [B] [N]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "tools", "ablate_libs", os.environ.get("STAMPS_LIB", "stamps") + ".so")
os.environ["SOT_LIB_PATH"] = lib
import torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
call = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
sets = [(torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)) for _ in range(6)]
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
plan = nat.PositionPlan(pos, pos2)
for i in range(300):
    if call == "area":
        nat.forward_rows(*sets[i % 6], pos, pos2, 1.0, 8, plan)
    elif call == "fwd":
        nat.forward_rows(*sets[i % 6], pos, pos2, 2.0, 15, plan)
    else:
        nat.loss_and_grad(*sets[i % 6], pos, pos2, 2.0, 15, plan)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 64)()
ctypes.CDLL(lib).sot_debug_read_stamps(out, 64)
common = {1: "wait for the row's loads, staging stores, chunk sums", 2: "barrier 1", 3: "(chunk sums from LDS: none with the column fetch)", 4: "(barrier 2: none with the column fetch)",
          5: "columns + fold (waves 0,1) / owner reads", 6: "barrier 3", 7: "division + fp64 accumulation + wave scans", 8: "barrier 4"}
names = dict(common)
if call == "area":
    names.update({9: "wave totals, CDF values, next loads issued", 10: "area + wave sum + barrier 5 + store"})
elif call == "fwd":
    names.update({9: "wave totals, CDF values to LDS, next loads issued, barrier 5", 10: "partition search", 11: "merge walk", 12: "wave sum + barrier 6 + store"})
else:
    names.update({9: "wave totals, CDF values to LDS, next loads issued, barrier 5", 10: "partition search + gradient walk", 11: "loss wave sum + barrier 6",
                  12: "gradient reads, fp64 reverse sums, scans, exchange (barrier 7)", 13: "output arithmetic, stores", 14: "last barrier"})
rows = out[0]
tot = sum(out[i] for i in names)
print(f"{call} {B} x {N}: {rows} rows of one workgroup: {tot / rows:.0f} cycles per row")
rec = {}
for i, n in names.items():
    print(f"  {i:2d} {n:64s} {out[i] / rows:8.0f} cycles  {100 * out[i] / tot:5.1f} %")
    rec[n] = out[i] / rows
print(json.dumps({"call": call, "B": B, "N": N, "cycles_per_row": tot / rows, "phases": rec}))
