#!/bin/bash
# LDS / VALU counters of the timing-only ablation builds (fab0 = full kernel, fab1 = no walk, fab2 = no search, fab3 = neither,
# fab4 = no row mass): attributes LDS cycles and bank conflicts to the phases.  Output: gpurun_out/pmc_ablate.txt
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ablate; mkdir -p $OUT; : > gpurun_out/pmc_ablate.txt
for v in fab0 fab1 fab2 fab3 fab4; do
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/$v -- python3 tools/pmc_variant.py $v > $OUT/$v.log 2>&1
  python3 - $OUT/$v $v >> gpurun_out/pmc_ablate.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "sot_forward_full_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v) / 8192, 1) for k, v in sorted(acc.items())}, "(per row)")
PY
done
cat gpurun_out/pmc_ablate.txt
