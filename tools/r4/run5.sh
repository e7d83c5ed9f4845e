#!/bin/bash
# round 4, GPU call 5: merge sort in LDS (per-row positions, segmented sort, position plans) + hoisted permutation lookups: full GPU suite + bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r4e/pytest_all.log 2>&1; echo "pytest rc=$?"
tail -n 6 gpurun_out/r4e/pytest_all.log
python bench.py > gpurun_out/r4e/bench.json 2> gpurun_out/r4e/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4e/bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step")}, r["roofline"]["frac"], r["roofline"]["kernel_ms"])
for k in ("paper_mode", "merge_p1", "training_form"):
    print(k, {a: r["roofline"][k].get(a) for a in ("kernel_ms", "frac", "stream_ms_per_call")})
for k, v in r["extras"].items():
    if isinstance(v, dict) and "ms" in v:
        print(f"{k:64s} {v['ms']*1e3:9.1f} us  frac {v.get('frac', 0):.3f}  host {v.get('host_us_per_call')}")
PY
python - <<'PY'
# segmented sort API: time torch.sort replacement on [4096, 2048] and [16384, 512] random keys
import torch, time
import sys; sys.path.insert(0, '.')
from sot_amd import _native as nat
dev = torch.device('cuda:0')
for B, N in ((4096, 2048), (16384, 512), (1024, 8192)):
    k = torch.rand(B, N, device=dev)
    for _ in range(5): v, i = nat.segmented_sort(k)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): v, i = nat.segmented_sort(k)
    e1.record(); torch.cuda.synchronize()
    tv, ti = torch.sort(k, dim=1, stable=True)
    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e2.record()
    for _ in range(20): tv, ti = torch.sort(k, dim=1, stable=True)
    e3.record(); torch.cuda.synchronize()
    print(f"segmented sort {B}x{N}: {e0.elapsed_time(e1)/20*1e3:.1f} us (torch.sort on the GPU {e2.elapsed_time(e3)/20*1e3:.1f} us), equal: {torch.equal(v, tv) and torch.equal(i, ti)}")
PY
