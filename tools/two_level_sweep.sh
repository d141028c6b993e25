#!/bin/bash
# Knobs of the path kernel on two-level trees (cloud-2000 / cloud-100000, 1920x1080, 4 spp).  Usage: tools/two_level_sweep.sh > out.txt
run() { echo "== $*"; env "$@" python3 tools/two_level_bench.py --render-only --structures two --spp 4 2>&1 | grep -v amdgpu.ids | python3 -c '
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.rstrip()); continue
    print("  ", d["scene"], d["spp4"])'; }
run HRT_DUMMY=1
for v in 32 40 48; do run HRT_REFILL_THRESHOLD=$v; done
for v in 0 25 60; do run HRT_POSTPONE_PCT=$v; done
run HRT_TRAVERSE_BLOCKS_PER_CU=12
run HRT_LIB=$PWD/nvidia-optix-ray-tracer_amd/lib/libhrt_w3.so HRT_TRAVERSE_BLOCKS_PER_CU=12
run HRT_LIB=$PWD/nvidia-optix-ray-tracer_amd/lib/libhrt_w3.so HRT_TRAVERSE_BLOCKS_PER_CU=12 HRT_REFILL_THRESHOLD=40
