#!/bin/bash
# opt-in merge-free p = 1 training form: its tests, then timings against the merge backward
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4v; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge_free_training_form or tie_free_gradient" > $O/pytest_area_train.log 2>&1; tail -n 12 $O/pytest_area_train.log | cut -c1-300
python - > $O/time.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device('cuda:0')
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for B, N in ((8192, 2048), (16384, 1025), (4096, 1025), (1024, 1025), (32768, 512), (65536, 257), (4096, 2049)):
    g = torch.Generator(device=dev).manual_seed(5)
    sets = [(torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)) for _ in range(4)]
    pos = torch.linspace(0, 1, N, device=dev); pos2 = pos.clone()
    plan = nat.PositionPlan(pos, pos2)
    i = [0]
    def lg(flags):
        i[0] += 1
        x, y = sets[i[0] % 4]
        nat.loss_and_grad(x, y, pos, pos2, 1.0, flags, plan)
    def fw(flags):
        i[0] += 1
        x, y = sets[i[0] % 4]
        nat.loss_fused(x, y, pos, pos2, 1.0, flags, plan)
    print(f"{B}x{N} p=1: forward+mean {timed(lambda: fw(8)):.1f} us; training form merge-free (opt-in) {timed(lambda: lg(8 | 256)):.1f} us, merge walk (default) {timed(lambda: lg(8)):.1f} us")
PY
cat $O/time.txt
