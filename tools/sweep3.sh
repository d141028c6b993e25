#!/bin/bash
OUT=${1:-gpurun_out/sweep3.txt}
: > "$OUT"
for fc in 16 32 64 128 256; do
 for ts in 0 1; do
   r=$(HRT_FETCH_CHUNK=$fc HRT_TAIL_SPLIT=$ts python3 bench.py --steps 2 --warmup 1 --spp 16 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms']['traverse'], d['kernel_ms']['traverse_any'], d['roofline']['avg_launch_ms'], d['roofline']['nodes_per_ray'])")
   echo "chunk=$fc split=$ts : $r" | tee -a "$OUT"
 done
done
