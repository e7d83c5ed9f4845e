#!/bin/bash
# usage: tools/knob_sweep.sh ENVVAR v1 v2 ... -- [bench args]
VAR=$1; shift; VALS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do VALS+=("$1"); shift; done; shift
for v in "${VALS[@]}"; do
  env $VAR=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | tail -1 > /tmp/k.json
  python - "$VAR" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/k.json")); print(sys.argv[1], sys.argv[2], "kernel_ms", round(d["roofline"]["kernel_ms"], 4), "ms_per_step", round(d["ms_per_step"], 4))
PY
done
