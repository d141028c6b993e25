#!/bin/bash
OUT=${1:-gpurun_out/sweep11.txt}
: > "$OUT"
for pp in 0 25 50 75; do
 for bpc in 16 20; do
   r=$(HRT_POSTPONE_PCT=$pp HRT_TRAVERSE_BLOCKS_PER_CU=$bpc python3 bench.py --steps 2 --warmup 1 --spp 16 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms']['traverse'], d['kernel_ms']['traverse_any'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")
   echo "postpone=$pp waves/cu=$bpc : $r" | tee -a "$OUT"
 done
done
