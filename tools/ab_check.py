"""Correctness spot check of diagnostic library variants (tools/ablate_libs/*.so) against the C oracle: B=48, N=2048 (AB_N=... for another length)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sys.argv[1:]:
    code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import os
os.environ['SOT_LIB_PATH'] = {ROOT!r} + '/tools/ablate_libs/' + {name!r} + '.so'
import numpy as np, torch, sot_amd
from sot_amd import _native as nat
nat.load(build_if_missing=False)
from oracle.inputs import gen_inputs
from oracle import sot_oracle as so
dev = torch.device('cuda:0')
worst = 0.0
for kind in ('peaky', 'uniform', 'dyadic'):
    N = {int(os.environ.get('AB_N', '2048'))}
    x, y = gen_inputs(kind, 48, N, N, 7)
    pos = torch.linspace(0, 1, N)
    for flags, p in ((0, 1.0), (15, 2.0), (1, 1.0)):
        got = nat.forward_rows(x.to(dev), y.to(dev), pos.to(dev), pos.to(dev).clone(), p, flags).cpu().numpy()
        want = so.forward(x.numpy(), y.numpy(), pos.numpy(), pos.numpy(), p=p, flags=flags)
        worst = max(worst, float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-30))))
print('max rel err vs oracle', worst)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print(f"{name:14s}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
