#!/bin/bash
# HRT_XCD_BANDS (each XCD's workgroups take a contiguous band of the tile first) against the default (every eighth slice): C4 and an 8 M soup.
# Usage (on the GPU box): tools/xcd_bands_sweep.sh > gpurun_out/xcd_bands.txt
set -u
for b in 0 1; do
  echo "== HRT_XCD_BANDS=$b bench.py (C4, 256 spp)"
  HRT_XCD_BANDS=$b timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-builder 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1
done
for b in 0 1; do
  echo "== HRT_XCD_BANDS=$b large_scene_bench.py --tris 8000000"
  HRT_XCD_BANDS=$b timeout -k 10 400 python3 tools/large_scene_bench.py --tris 8000000 --no-count 2>&1 | tail -1 || exit 1
done
