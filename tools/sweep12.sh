#!/bin/bash
OUT=${1:-gpurun_out/sweep12.txt}
: > "$OUT"
for fc in 16 32 64 128; do
 for rt in 8 16 24; do
   r=$(HRT_FETCH_CHUNK=$fc HRT_REFILL_THRESHOLD=$rt python3 bench.py --steps 2 --warmup 1 --spp 32 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")
   echo "chunk=$fc refill=$rt : $r" | tee -a "$OUT"
 done
done
