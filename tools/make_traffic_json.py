"""Turns the FETCH_SIZE / WRITE_SIZE PMC passes of tools/profile_round.sh into profiles/<tag>_hbm_traffic.json
(what bench.py reports as roofline.traffic).  Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM:
FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of a 16-B-per-lane
coalesced stream (our staging loads), so it is doubled; WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import sys

prof_dir, workload, out = sys.argv[1], sys.argv[2], sys.argv[3]
KERNEL = sys.argv[4] if len(sys.argv) > 4 else "sot_forward_full_kernel<256, 8, 1, 1, false, false, 0>"  # the bench workload's variant


def mean_counter(sub, counter, kernel_substr):
    vals = []
    for f in glob.glob(os.path.join(prof_dir, sub, "**/*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


fetch_kib, nf = mean_counter("pmc_fetch", "FETCH_SIZE", KERNEL)
write_kib, nw = mean_counter("pmc_write", "WRITE_SIZE", KERNEL)
rec = {"workload": workload, "kernel": KERNEL, "fetch_size_kib_raw": fetch_kib, "write_size_kib": write_kib,
       "dispatches": [nf, nw], "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B on wide coalesced streams; round 4 fetches whole 128-B lines with dword loads: the raw counter again reads half of the bytes the kernel provably loads)",
       "hbm_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
       "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), {os.path.basename(prof_dir)}"}
json.dump(rec, open(out, "w"), indent=1)
print(rec)
