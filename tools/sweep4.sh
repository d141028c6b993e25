#!/bin/bash
OUT=${1:-gpurun_out/sweep4.txt}
: > "$OUT"
for pp in 0 10 25 40 60 80; do
   r=$(HRT_POSTPONE_PCT=$pp python3 bench.py --steps 2 --warmup 1 --spp 16 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms']['traverse'], d['kernel_ms']['traverse_any'], d['roofline']['avg_launch_ms'], d['roofline']['nodes_per_ray'], d['roofline']['prims_per_ray'], d['lanes'])")
   echo "postpone=$pp : $r" | tee -a "$OUT"
done
