#!/bin/bash
OUT=${1:-gpurun_out/sweep10.txt}
: > "$OUT"
for cfg in "80 48" "128 48" "80 64" "128 64" "96 48"; do
 set -- $cfg
   r=$(HRT_NODE_STRIDE=$1 HRT_PRIM_STRIDE=$2 python3 bench.py --steps 2 --warmup 1 --spp 16 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms']['traverse'], d['kernel_ms']['traverse_any'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")
   echo "node_stride=$1 prim_stride=$2 : $r" | tee -a "$OUT"
done
