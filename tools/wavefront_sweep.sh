#!/bin/bash
# wavefront mode (HRT_FUSED=0) on C4 at 16 spp: traverse kernel (lean = k_trace_queue / round 1's k_traverse) x hipGraph replay.
# Usage: tools/wavefront_sweep.sh <outdir>
OUT=${1:-gpurun_out/wavefront}; mkdir -p $OUT
for cfg in "1 0" "1 1" "0 0" "0 1"; do set -- $cfg
  HRT_FUSED=0 HRT_WAVEFRONT_LEAN=$1 HRT_WAVEFRONT_GRAPH=$2 HRT_BENCH_NO_TIMING=1 python3 bench.py --steps 3 --warmup 1 --spp 16 --no-alt-builder --cpu-seconds 6 > $OUT/wf_$1_$2.json 2> $OUT/wf_$1_$2.err
  python3 -c "
import json
try:
    d=json.loads(open('$OUT/wf_$1_$2.json').read().strip().splitlines()[-1]); print('wavefront lean=$1 graph=$2 :', d['value'], 'Mrays/s', d['ms_per_step'], 'ms; parity', d.get('parity', {}).get('linear_radiance_bit_exact'))
except Exception as e: print('wavefront lean=$1 graph=$2 FAILED', e)
" | tee -a $OUT/wavefront.txt
done
