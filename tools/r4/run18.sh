#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4p; rm -rf $O; mkdir -p $O
python tools/r4/sort_probe.py > $O/sort.txt 2>&1; tail -n 2 $O/sort.txt
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
timeout 900 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -n 5 $O/bench.err
