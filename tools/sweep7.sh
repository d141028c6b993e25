#!/bin/bash
OUT=${1:-gpurun_out/sweep7.txt}
: > "$OUT"
for bpc in 6 8 10 11; do
 for rt in 8 16 24; do
   r=$(HRT_TRAVERSE_BLOCKS_PER_CU=$bpc HRT_REFILL_THRESHOLD=$rt python3 bench.py --steps 2 --warmup 1 --spp 16 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms']['traverse'], d['kernel_ms']['traverse_any'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")
   echo "bpc=$bpc refill=$rt : $r" | tee -a "$OUT"
 done
done
