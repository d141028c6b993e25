#!/bin/bash
# HRT_LEAF_HOLD: how many leaf groups a lane may queue before its node work waits for primitive tests (4 = the stack's depth: only when it is full).
# Fewer = leaves tested sooner after their node is visited = hits found earlier cull more, against less overlap of node and primitive work.
# Usage (GPU box): tools/leaf_hold_sweep.sh > gpurun_out/leaf_hold.txt
set -u
for h in ${1:-4 3 2 1}; do   # (0 = the default: by scene)
  echo "== HRT_LEAF_HOLD=$h"
  HRT_LEAF_HOLD=$h timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-builder 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4', d['value'], 'Mrays/s', d['ms_per_step'], 'ms')" || exit 1
  HRT_LEAF_HOLD=$h timeout -k 10 300 python3 tools/two_level_bench.py --render-only --spp 4 --particles 2000 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['scene'], d['structure'], d['spp4']['Mrays_per_s'], 'Mrays/s')" || exit 1
  HRT_LEAF_HOLD=$h timeout -k 10 300 python3 tools/two_level_bench.py --render-only --spp 4 --particles 100000 --structures two 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['scene'], d['structure'], d['spp4']['Mrays_per_s'], 'Mrays/s')" || exit 1
  HRT_LEAF_HOLD=$h timeout -k 10 300 python3 tools/two_level_bench.py --render-only --spp 4 --particles 2000 --structures flat --scene column 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['scene'], d['structure'], d['spp4']['Mrays_per_s'], 'Mrays/s')" || exit 1
  for i in 1 2; do HRT_LEAF_HOLD=$h nvidia-optix-ray-tracer_amd/lib/hrt_time_render tests/golden/files/config.json tests/golden/files -1 /tmp/o.ppm 2>&1 | tail -1; done
done
