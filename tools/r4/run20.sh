#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4r; rm -rf $O; mkdir -p $O
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES"; do
  d=$O/pmc_$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 tools/r4/sort_probe.py > $d.log 2>&1
done
python3 - $O > $O/pmc.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "segmented_sort" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, sum(v) / len(v), len(v))
PY
cat $O/pmc.txt
python - > $O/perrow.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(5)
B, N = 4096, 2048
x, y = torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)
px, py = torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
print("per-row positions forward 4096x2048 (paper mode): %.1f us" % timed(lambda: nat.forward_rows(x, y, px, py, 2.0, 15, None)))
sx, sy = torch.sort(px, 1).values, torch.sort(py, 1).values
print("  rows already sorted: %.1f us" % timed(lambda: nat.forward_rows(x, y, sx, sy, 2.0, 15, None)))
one = torch.ones(1, device=dev)
print("per-row positions backward (both gradients): %.1f us" % timed(lambda: nat.backward_rows(x, y, px, py, 2.0, 15, one)))
print("per-row position gradients: %.1f us" % timed(lambda: nat.position_grads(x, y, px, py, 2.0, 15, one)))
PY
cat $O/perrow.txt
