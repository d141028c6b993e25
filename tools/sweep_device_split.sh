#!/bin/bash
# tools/sweep_device_split.sh OUT [VAR=values ...] -- C4 bench with the device's spatial-split builder under a few settings
out=$1; shift
mkdir -p "$(dirname "$out")"; : > "$out"
run() {
  echo "== $*" >> "$out"
  env HRT_FAST_TRACE_BUILD=device "$@" timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-alt-builder 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); c=d["config"]; print(d["value"], d["ms_per_step"], "nodes", c["bvh_nodes"], "bytes", c["bvh_bytes"], "build_s", c["bvh_build_s"])
' >> "$out" 2>&1
}
for spec in "$@"; do
  run $spec
done
cat "$out"
