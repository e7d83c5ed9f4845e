#!/bin/bash
# Occupancy-throttle experiment: pad the forward kernel's LDS request and time it.
for x in 0 8000 20000 48000; do
  SOT_DEBUG_EXTRA_LDS=$x python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | tail -1 > /tmp/occ.json
  python - "$x" <<'PY'
import json, sys
d = json.load(open("/tmp/occ.json"))
print("extra_lds", sys.argv[1], "kernel_ms", round(d["roofline"]["kernel_ms"], 4), "frac", round(d["roofline"]["frac"], 4))
PY
done
