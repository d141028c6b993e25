#!/bin/bash
# The regeneration threshold (lanes that wait before a wave regenerates) on the particle clouds, flattened and two-level, and the particle column.
# Usage (GPU box): tools/refill_sweep_clouds.sh "20 24 28 32" > gpurun_out/refill_clouds.txt
for v in ${1:-8 12 16 20}; do echo "== HRT_REFILL_THRESHOLD=$v"
  HRT_REFILL_THRESHOLD=$v python3 tools/two_level_bench.py --render-only --spp 4 ${2:-} 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  ', d['scene'], d['structure'], d['spp4']['Mrays_per_s'])"
  HRT_REFILL_THRESHOLD=$v python3 tools/two_level_bench.py --render-only --spp 4 --particles 2000 --structures flat --scene column 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  ', d['scene'], d['structure'], d['spp4']['Mrays_per_s'])"
done
