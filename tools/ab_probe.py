"""Interleaved A/B timing of diagnostic library variants (tools/ablate_libs/*.so): forward kernel, B=8192, N=2048."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:]
mode_flags = os.environ.get("AB_FLAGS", "8")   # 8 = REQUIRE_SORT only (p1); 15 = paper cutoff with p=2
pval = os.environ.get("AB_P", "1.0")
for rnd in range(3):
    for name in names:
        code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import os
os.environ['SOT_LIB_PATH'] = {ROOT!r} + '/tools/ablate_libs/' + {name!r} + '.so'
import sot_amd, torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device('cuda:0'); B, N = {int(os.environ.get('AB_B', '8192'))}, {int(os.environ.get('AB_N', '2048'))}
g = torch.Generator(device=dev).manual_seed(0)
sets = [(torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)) for _ in range({int(os.environ.get('AB_SETS', '6'))})]
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
plan = nat.PositionPlan(pos, pos2)
call = {os.environ.get('AB_CALL', 'fwd')!r}
one = torch.ones(1, device=dev)
def run(i):
    x, y = sets[i % len(sets)]
    if call == 'fwd': nat.forward_rows(x, y, pos, pos2, {pval}, {mode_flags}, plan)
    elif call == 'lg': nat.loss_and_grad(x, y, pos, pos2, {pval}, {mode_flags}, plan)
    else: nat.backward_rows(x, y, pos, pos2, {pval}, {mode_flags}, one, need_gx=(call == 'bwdxy'), plan=plan, grad_scale=1.0 / B)
for i in range(500): run(i)
res = []
for rep in range(3):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(200): run(i)
    b.record(); torch.cuda.synchronize()
    res.append(round(a.elapsed_time(b) * 5, 1))
print(res)
"""
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        print(f"{name:14s}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
