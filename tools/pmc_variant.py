"""Runs one kernel of a diagnostic library variant (tools/ablate_libs/<name>.so, or the product library for name '-') 60 times
(for `rocprofv3 --pmc ... -- python3 tools/pmc_variant.py <name> [flags] [p] [call] [B] [N]`); call: fwd | lg (loss_and_grad) | bwd."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] != "-":
    os.environ["SOT_LIB_PATH"] = os.path.join(ROOT, "tools", "ablate_libs", sys.argv[1] + ".so")
import torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
arg = lambda i, d: sys.argv[i] if len(sys.argv) > i else d  # noqa: E731
flags, p, call, B, N = int(arg(2, "8")), float(arg(3, "1.0")), arg(4, "fwd"), int(arg(5, "8192")), int(arg(6, "2048"))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
nsets = max(2, min(6, (800 << 20) // (8 * B * N)))
sets = [(torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)) for _ in range(nsets)]
pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
plan = nat.PositionPlan(pos, pos2)
one = torch.ones(1, device=dev)
for i in range(60):
    x, y = sets[i % nsets]
    if call == "fwd":
        nat.forward_rows(x, y, pos, pos2, p, flags, plan)
    elif call == "lg":
        nat.loss_and_grad(x, y, pos, pos2, p, flags, plan)
    else:
        nat.backward_rows(x, y, pos, pos2, p, flags, one, need_gx=(call == "bwdxy"), plan=plan, grad_scale=1.0 / B)
torch.cuda.synchronize()
