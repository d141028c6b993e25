#!/bin/bash
OUT=${1:-gpurun_out/sweep9.txt}
: > "$OUT"
for cfg in "0 14" "0 16" "0 18" "0 20" "1 11"; do
 set -- $cfg
 for rt in 8 16; do
   r=$(HRT_LDS_GATHER=$1 HRT_TRAVERSE_BLOCKS_PER_CU=$2 HRT_REFILL_THRESHOLD=$rt python3 bench.py --steps 2 --warmup 1 --spp 16 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms']['traverse'], d['kernel_ms']['traverse_any'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")
   echo "dma=$1 waves/cu=$2 refill=$rt : $r" | tee -a "$OUT"
 done
done
