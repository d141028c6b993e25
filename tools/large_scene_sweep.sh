#!/bin/bash
# Out of the Infinity Cache, bytes per visit are the lever: record / node strides and the builders at 8 M and 32 M triangles, one scene
# generation per size.  Usage: tools/large_scene_sweep.sh "8000000 32000000" > profiles/rNN_large_scenes_sweep.txt
SIZES=${1:-"8000000 32000000"}
run() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ("scene","builder","mrays_per_s","kernel_ms_per_step","bvh_nodes","bvh_depth","bvh_alloc_bytes","nodes_per_ray","prims_per_ray") if k in d})'; }
for N in $SIZES; do
  run python3 tools/large_scene_bench.py --tris $N
  run HRT_PRIM_STRIDE=48 python3 tools/large_scene_bench.py --tris $N
  run HRT_PRIM_STRIDE=48 HRT_NODE_STRIDE=80 python3 tools/large_scene_bench.py --tris $N
  run python3 tools/large_scene_bench.py --tris $N --fast-trace
  run HRT_PRIM_STRIDE=48 python3 tools/large_scene_bench.py --tris $N --fast-trace
done
