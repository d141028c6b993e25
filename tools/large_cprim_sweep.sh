#!/bin/bash
# Out of the Infinity Cache: does the collapse's primitive cost (HRT_BVH_CPRIM: leaf size against node count) want another value than C4's 0.45?
# Usage (GPU box): tools/large_cprim_sweep.sh "8000000 32000000" > gpurun_out/large_cprim.txt
set -u
for N in ${1:-8000000}; do for FT in "" "--fast-trace"; do for C in 0.25 0.45 0.8 1.5; do
  echo "== tris $N $FT HRT_BVH_CPRIM=$C"
  HRT_BVH_CPRIM=$C timeout -k 10 400 python3 tools/large_scene_bench.py --tris $N --no-count $FT 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['mrays_per_s'], 'Mrays/s', d['ms_per_step'], 'ms; nodes', d['bvh_nodes'], 'depth', d['bvh_depth'], 'alloc MB', round(d['bvh_alloc_bytes'] / 1e6))" || exit 1
done; done; done
