#!/bin/bash
OUT=${1:-gpurun_out/sweep2.txt}
: > "$OUT"
for ss in 1 2 3 4 6; do
 for bpc in 6; do
   r=$(HRT_SUBSTREAMS=$ss HRT_TRAVERSE_BLOCKS_PER_CU=$bpc python3 bench.py --steps 2 --warmup 1 --spp 16 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms']['traverse'], d['kernel_ms']['traverse_any'], d['roofline']['frac'])")
   echo "substreams=$ss bpc=$bpc : $r" | tee -a "$OUT"
 done
done
