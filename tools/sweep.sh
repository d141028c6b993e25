#!/bin/bash
# Sweep traversal tuning knobs (env vars read at context creation) with short bench runs.
# Usage: tools/sweep.sh <outfile> ; prints "<knobs> value ms_traverse"
OUT=${1:-gpurun_out/sweep.txt}
: > "$OUT"
for pp in 0 15 25 40 60; do
 for rt in 8 16 24 32; do
  for bpc in 6; do
   r=$(HRT_POSTPONE_PCT=$pp HRT_REFILL_THRESHOLD=$rt HRT_TRAVERSE_BLOCKS_PER_CU=$bpc python3 bench.py --steps 2 --warmup 1 --spp 8 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms']['traverse'], d['kernel_ms']['traverse_any'], d['roofline']['nodes_per_ray'], d['roofline']['prims_per_ray'], d['roofline']['frac'])")
   echo "postpone=$pp refill=$rt bpc=$bpc : $r" | tee -a "$OUT"
  done
 done
done
