"""Runs the forward kernel of one diagnostic library variant (tools/ablate_libs/<name>.so) 60 times on B=8192, N=2048
(for `rocprofv3 --pmc ... -- python3 tools/pmc_variant.py <name> [flags] [p]`); with a position plan (the module's path)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SOT_LIB_PATH"] = os.path.join(ROOT, "tools", "ablate_libs", sys.argv[1] + ".so")
import torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 8
p = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
dev = torch.device("cuda:0"); B, N = 8192, 2048
g = torch.Generator(device=dev).manual_seed(0)
sets = [(torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)) for _ in range(6)]
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
plan = nat.PositionPlan(pos, pos2)
for i in range(60):
    nat.forward_rows(*sets[i % 6], pos, pos2, p, flags, plan)
torch.cuda.synchronize()
