#!/bin/bash
# round 4, GPU call 3: full GPU test suite (every row against the OpenMP oracle), phase stamps, bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r4c/pytest_all.log 2>&1; echo "pytest rc=$?"
tail -n 5 gpurun_out/r4c/pytest_all.log
for c in "fwd 8192 2048" "area 8192 2048" "lg 8192 2048" "lg 16384 1025" "fwd 16384 1025"; do
  python tools/r4/phase_stamps.py $c >> gpurun_out/r4c/stamps.txt 2>&1
done
cat gpurun_out/r4c/stamps.txt
python bench.py > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4c/bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step")}, r["roofline"]["frac"], r["roofline"]["kernel_ms"])
for k in ("paper_mode", "merge_p1", "training_form"):
    print(k, {a: r["roofline"][k].get(a) for a in ("kernel_ms", "frac", "stream_ms_per_call")})
for k, v in r["extras"].items():
    if isinstance(v, dict) and "ms" in v:
        print(f"{k:64s} {v['ms']*1e3:9.1f} us  frac {v.get('frac', 0):.3f}  host {v.get('host_us_per_call')}")
print(r["cpu_baseline"])
PY
