"""Two diagnostic library variants give the same training-form results (loss rows + gradient) bit for bit:
    AB_N=2000 AB_B=512 python tools/ab_equal.py nameA nameB      (paper mode, p = 2)"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
outs = []
for name in sys.argv[1:3]:
    f = tempfile.NamedTemporaryFile(suffix=".pt", delete=False).name
    code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import os
os.environ['SOT_LIB_PATH'] = {ROOT!r} + '/tools/ablate_libs/' + {name!r} + '.so'
import torch, sot_amd
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device('cuda:0'); B, N = {int(os.environ.get('AB_B', '512'))}, {int(os.environ.get('AB_N', '2000'))}
g = torch.Generator(device=dev).manual_seed(3)
x, y = torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
res = []
for flags, p in ((15, 2.0), (8, 1.0), (9, 3.0)):
    out = nat.loss_and_grad(x, y, pos, pos2, p, flags, nat.PositionPlan(pos, pos2))
    res.append([t.cpu() for t in out if torch.is_tensor(t)])
torch.save(res, {f!r})
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if r.returncode:
        print(name, r.stderr[-600:]); sys.exit(1)
    outs.append(f)
import torch
a, b = torch.load(outs[0]), torch.load(outs[1])
for k, (ra, rb) in enumerate(zip(a, b)):
    print("mode", k, [bool(torch.equal(u, v)) for u, v in zip(ra, rb)], [tuple(u.shape) for u in ra])
