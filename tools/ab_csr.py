import os, sys, subprocess
ROOT='/root/repo'
for rnd in range(2):
    for name in sys.argv[1:]:
        code=f"""
import os, sys; sys.path.insert(0, {ROOT!r})
os.environ['SOT_LIB_PATH'] = {ROOT!r} + '/tools/ablate_libs/' + {name!r} + '.so'
import torch
from sot_amd import _native as nat
from sot_amd.bench_inputs import ragged_supports
from sot_amd.losses import wasserstein_1d_csr
nat.load(build_if_missing=False)
dev = torch.device('cuda:0')
rs = ragged_supports(8192, 512, 1234)
(xw, xp, xo), (yw, yp, yo) = [[t.to(dev) for t in part] for part in rs['csr']]
kw = dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True)
def run(): wasserstein_1d_csr(xw, xp, xo, yw, yp, yo, rs['max_n'], rs['max_m'], **kw)
for _ in range(50): run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(200): run()
b.record(); torch.cuda.synchronize()
print(round(a.elapsed_time(b) * 5, 1))
"""
        r=subprocess.run([sys.executable,'-c',code],capture_output=True,text=True)
        print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
