#!/bin/bash
OUT=${1:-gpurun_out/sweep8.txt}
: > "$OUT"
for bpc in 11 12 13 14; do
 for fc in 32 64; do
   r=$(HRT_TRAVERSE_BLOCKS_PER_CU=$bpc HRT_REFILL_THRESHOLD=8 HRT_FETCH_CHUNK=$fc python3 bench.py --steps 2 --warmup 1 --spp 16 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms']['traverse'], d['kernel_ms']['traverse_any'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")
   echo "bpc=$bpc chunk=$fc : $r" | tee -a "$OUT"
 done
done
