"""Timing-only ablation of the forward kernel at full occupancy (diagnostic; results are wrong on purpose).
Build (CPU container):  python tools/ablate.py build      Run (GPU box):  python tools/ablate.py run"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {0: "full kernel", 1: "no merge walk", 2: "no partition search", 3: "no walk + no search", 4: "no row mass",
            8: "no division", 16: "no wave scan", 31: "staging + barriers only"}
LIBDIR = os.path.join(ROOT, "tools", "ablate_libs")
if sys.argv[1] == "build":
    import sot_amd
    from concurrent.futures import ThreadPoolExecutor
    def one(v):
        out = os.path.join(LIBDIR, f"libsot_ablate_{v}.so")
        subprocess.run([sot_amd.build.hipcc_path(), *sot_amd.build.HIPCC_FLAGS, "-shared", f"-DSOT_ABLATE={v}", "-DSOT_PART=145", "-DSOT_STUB_MISSING_PARTS", "-o", out,
                        sot_amd.build.SRC], check=True)
        return out
    with ThreadPoolExecutor(4) as ex:
        print(list(ex.map(one, VARIANTS)))
else:
    for v, name in VARIANTS.items():
        code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import sot_amd, torch
sot_amd.build.LIB = {os.path.join(LIBDIR, f'libsot_ablate_{v}.so')!r}
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device('cuda:0'); B, N = 8192, 2048
g = torch.Generator(device=dev).manual_seed(0)
sets = [(torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)) for _ in range(6)]
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
plan = nat.PositionPlan(pos, pos2)
for i in range(10): nat.forward_rows(*sets[i % 6], pos, pos2, 1.0, 8, plan)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(100): nat.forward_rows(*sets[i % 6], pos, pos2, 1.0, 8, plan)
b.record(); torch.cuda.synchronize()
print(round(a.elapsed_time(b) * 10, 1))
"""
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        print(f"ablate {v:2d} {name:28s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]} us")
