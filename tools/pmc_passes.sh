#!/bin/bash
# Separate rocprofv3 --pmc passes (SQ issue / LDS sets, never combined with a trace domain other than --kernel-trace) of ONE python
# program, averaged per kernel whose name contains <match>.  Replaces the per-call copies of this loop that round 4 kept as
# tools/r4/run18.sh / run20.sh.
# Usage (GPU box): tools/pmc_passes.sh <out tag> <kernel-name match> <script.py> [script args...]   -> gpurun_out/<tag>/pmc.txt
set -eu
cd "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
export TMPDIR=/tmp
O="gpurun_out/${1:?output tag}"; MATCH="${2:?kernel-name match}"; shift 2
mkdir -p "$O"
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1)); d="$O/pmc_$i"; rm -rf -- "$d"
  # shellcheck disable=SC2086  (the counter set is a word list on purpose)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$d" -- python3 "$@" > "$d.log" 2>&1 || echo "pass $i failed (see $d.log)"
done
python3 - "$O" "$MATCH" > "$O/pmc.txt" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0][:110]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {sum(v) / len(v):16.1f}   ({len(v)} dispatches)")
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:   # KiB units; gfx950: FETCH_SIZE x 2 for wide coalesced streams (MI355X_MICROARCH.md, HBM section)
        f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024 * 2; w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024
        print(f"    HBM bytes per dispatch: reads {f / 1e6:.2f} MB (FETCH_SIZE x 2), writes {w / 1e6:.2f} MB")
PY
cat "$O/pmc.txt"
