#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one kernel of library variants: tools/r4/traffic_pass.sh <out> <variant>:<flags>:<p>:<call>:<B>:<N> ...
# (separate --pmc passes, MI355X_MICROARCH.md "HBM"; FETCH_SIZE x 2 for 16-B-per-lane / wide coalesced streams on gfx950)
set -eu
cd "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
export TMPDIR=/tmp
OUT="gpurun_out/${1:?output tag}"; shift
rm -rf -- "$OUT"; mkdir -p "$OUT"; : > "$OUT.txt"
for spec in "$@"; do
  IFS=: read v flags p call B N <<< "$spec"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${v}_${call}_${B}_${N}_$c -- python3 tools/pmc_variant.py $v $flags $p $call $B $N > $OUT/$v.log 2>&1
  done
  python3 - $OUT "$v" "$call" "$B" "$N" >> $OUT.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
out, v, call, B, N = sys.argv[1:6]
B, N = int(B), int(N)
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{out}/{v}_{call}_{B}_{N}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "sot_" in k and "prepare" not in k and "reduce_mean" not in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
alg = {"fwd": B * (8 * N + 4), "lg": B * (12 * N + 4), "bwd": B * (12 * N), "bwdxy": B * (16 * N)}[call]
for k, d in acc.items():
    f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])) * 1024 * 2
    w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) * 1024
    print(f"{v} {call} {B}x{N} {k[:90]}: FETCH x2 {f/1e6:.2f} MB, WRITE {w/1e6:.2f} MB, sum {(f+w)/1e6:.2f} MB = {(f+w)/alg:.3f} x algorithmic {alg/1e6:.2f} MB "
          f"(reads {f/(B*8*N):.3f} x, writes {w/max(1, alg - B*8*N):.3f} x); dispatches {len(d['FETCH_SIZE'])}/{len(d['WRITE_SIZE'])}")
PY
done
cat $OUT.txt
