#!/bin/bash
# Round 5's GPU-box calls, one parametrised script: gpurun -- tools/gpu_call.sh <case> [args...]
# Every case writes under gpurun_out/<tag>/ (scratch); summaries worth judging are copied to profiles/ by hand.
set -u
cd "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
export TMPDIR=/tmp
CASE="${1:?case}"; shift
kstats() {  # kstats <outdir> <python script + args...>: rocprofv3 kernel-trace stats of one python program, top kernels printed
  local d="$1"; shift
  rm -rf -- "$d"; mkdir -p "$d"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$@" > "$d.log" 2>&1
  python3 - "$d" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
print(f"{'kernel':100s} {'calls':>6s} {'avg_us':>9s} {'total_us':>10s} {'pct':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:9.2f} {float(r['TotalDurationNs']) / 1e3:10.1f} {100 * float(r['TotalDurationNs']) / tot:6.2f}")
print(f"total kernel time {tot / 1e3:.1f} us over {sum(int(r['Calls']) for r in rows)} dispatches")
PY
}
case "$CASE" in
  paper_step_profile)   # <tag> [clips...]: kernel stats of the paper's loss step, MSSLoss alone, the SOT slice alone
    O="gpurun_out/${1:-r5a}"; shift || true
    mkdir -p "$O"
    for clips in ${@:-64 256}; do
      for what in full modules mss sot; do
        echo "== $what, $clips clips (45 steps incl. 5 warm-up)" | tee -a "$O/summary.txt"
        kstats "$O/${what}_${clips}" tools/r5/paper_step_probe.py "$clips" 40 "$what" | tee -a "$O/summary.txt"
        tail -n 1 "$O/${what}_${clips}.log" | tee -a "$O/summary.txt"
      done
    done ;;
  bench)                # <tag> [bench args]: the bench line
    O="gpurun_out/${1:-r5}"; shift || true
    mkdir -p "$O"
    timeout 1500 python bench.py "$@" > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
    tail -n 3 "$O/bench.err"
    python3 - "$O/bench.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step")}, "frac", r["roofline"]["frac"])
for k, v in r["extras"].items():
    if isinstance(v, dict) and "ms" in v and ("paper_loss" in k or "mssloss" in k or "sot_slice" in k or "per_row" in k or "config5" in k or "b1024" in k):
        print(f"{k:70s} {1e3 * v['ms']:9.1f} us")
PY
    ;;
  tests)                # <tag> [pytest args]: the GPU suite (or a selection)
    O="gpurun_out/${1:-r5}"; shift || true
    mkdir -p "$O"
    if [ $# -eq 0 ]; then set -- tests; fi     # default: the whole suite; else the given files / -k selections
    timeout 1700 python -m pytest -q -m gpu "$@" > "$O/pytest.log" 2>&1; echo "pytest rc=$?"
    tail -n 15 "$O/pytest.log" | cut -c1-300 ;;
  smoke)
    python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3 ;;
  kstats)               # <tag> <script.py> [args]: rocprofv3 kernel stats of one python program
    O="gpurun_out/${1:?tag}"; shift
    kstats "$O" "$@" ;;
  run)                  # <tag> <script.py> [args]: one python program, output kept
    O="gpurun_out/${1:?tag}"; shift
    mkdir -p "$O"
    timeout 1500 python3 "$@" > "$O/out.txt" 2>&1; echo "rc=$?"; tail -n 60 "$O/out.txt" | cut -c1-400 ;;
  *) echo "unknown case $CASE" >&2; exit 2 ;;
esac
