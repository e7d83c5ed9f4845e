#!/bin/bash
# SQ counters of the STFT forward kernel per frame: tools/pmc_stft.sh [n_fft hop]  -> gpurun_out/pmc_stft.txt
export TMPDIR=/tmp
OUT=gpurun_out/pmc_stft; rm -rf $OUT; mkdir -p $OUT
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_WAVES" \
            "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -- python3 tools/stft_only.py "$@" > $OUT/p$i.log 2>&1
done
python3 - $OUT > gpurun_out/pmc_stft.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "stft" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k[:80])
    for c, v in sorted(d.items()):
        print(f"   {c:26s} {sum(v) / len(v):16.0f}  per launch   {sum(v) / len(v) / 4096:12.1f} per frame")
PY
cat gpurun_out/pmc_stft.txt
