"""Merge forward of a library variant against the C oracle and against a second variant, several modes: python tools/ab_fwd_check.py A B  (AB_N, AB_B)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
outs = []
for name in sys.argv[1:3]:
    f = tempfile.NamedTemporaryFile(suffix=".pt", delete=False).name
    code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import os
os.environ['SOT_LIB_PATH'] = {ROOT!r} + '/tools/ablate_libs/' + {name!r} + '.so'
import numpy as np, torch, sot_amd
from sot_amd import _native as nat
nat.load(build_if_missing=False)
from oracle.inputs import gen_inputs
from oracle import sot_oracle as so
dev = torch.device('cuda:0'); B, N = {int(os.environ.get('AB_B', '777'))}, {int(os.environ.get('AB_N', '257'))}
res = []
for kind in ('peaky', 'dyadic', 'edge'):
    x, y = gen_inputs(kind, B, N, N, 13)
    g = torch.Generator().manual_seed(6)
    for posk in ('linspace', 'unsorted'):
        pos = torch.linspace(0, 1, N) if posk == 'linspace' else torch.rand(N, generator=g)
        pd, pd2 = pos.to(dev), pos.to(dev).clone()
        plan = nat.PositionPlan(pd, pd2)
        for flags, p in ((8, 1.0), (15, 2.0), (9, 2.0), (10, 1.0), (12, 3.0), (8, 1.5)):
            got = nat.forward_rows(x.to(dev), y.to(dev), pd, pd2, p, flags | nat.FLAG_NO_AREA, plan).cpu()
            want = so.forward(x.numpy(), y.numpy(), pos.numpy(), pos.numpy(), p=p, flags=flags & 15)
            err = float(np.max(np.abs(got.numpy() - want) / np.maximum(np.abs(want), 1e-6)))
            res.append((kind, posk, flags, p, err, got))
torch.save(res, {f!r})
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if r.returncode:
        print(name, r.stderr[-800:]); sys.exit(1)
    outs.append(f)
import torch
a = torch.load(outs[0]); b = torch.load(outs[1]) if len(outs) > 1 else a
worst = 0.0
for (k, pk, fl, p, ea, ga), (_, _, _, _, eb, gb) in zip(a, b):
    d = float(((ga - gb).abs() / gb.abs().clamp_min(1e-6)).max())
    worst = max(worst, ea)
    print(f"{k:8s} {pk:9s} flags {fl:3d} p {p}: err vs oracle {ea:.2e} | {eb:.2e}; A vs B {d:.2e}")
print("worst A vs oracle", worst)
