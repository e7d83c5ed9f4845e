"""Summarise a tools/profile_round.sh output directory: per-kernel stats + per-kernel mean PMC values."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


for f in find("stats_extras/**/*kernel_stats.csv"):
    print("== kernel stats of the WHOLE bench command, extras included (rocprofv3 --kernel-trace --stats; library kernels only) ==")
    for row in csv.DictReader(open(f)):
        name = row.get("Name", "")
        if "sot" in name:
            print(f"{name[:110]:110s} calls={row.get('Calls')} avg_ns={row.get('AverageNs')} total_ns={row.get('TotalDurationNs')}")

print("== kernel stats of the headline command (rocprofv3 --kernel-trace --stats) ==")
for f in find("stats/**/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        name = row.get("Name", "")[:90]
        print(f"{name:90s} calls={row.get('Calls')} avg_ns={row.get('AverageNs')} total_ns={row.get('TotalDurationNs')} pct={row.get('Percentage')}")

# per-dispatch durations from the kernel trace: the first few hundred launches of a process run ~10 % slower (clock ramp),
# so the steady-state average (last 30 dispatches = bench.py's timed region in tools/profile_round.sh) is listed as well
for f in find("stats/**/*kernel_trace.csv"):
    durs = defaultdict(list)
    for row in csv.DictReader(open(f)):
        durs[row.get("Kernel_Name", "")[:90]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print("== steady state (kernel trace, last 30 dispatches of each kernel with >= 60 dispatches) ==")
    for k, v in durs.items():
        if len(v) >= 60:
            tail = v[-30:]
            print(f"{k:90s} calls={len(v)} avg_ns_all={sum(v) / len(v):.1f} avg_ns_last30={sum(tail) / len(tail):.1f} min_ns={min(v)}")

for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:60]
            acc[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
    print(f"== {os.path.basename(d)} (mean per dispatch) ==")
    for k, cs in acc.items():
        if "sot_" not in k:
            continue
        print("  " + k)
        for c, v in sorted(cs.items()):
            print(f"     {c:28s} {sum(v) / len(v):16.1f}  (n={len(v)})")
