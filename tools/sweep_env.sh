#!/bin/bash
# Sweep tuning knobs (environment variables read at context creation) with short bench runs.
#   tools/sweep_env.sh <outfile> <bench args> -- VAR1=a,b,c [VAR2=x,y ...]
# e.g. tools/sweep_env.sh gpurun_out/s.txt --steps 2 --warmup 1 --spp 32 -- HRT_FETCH_CHUNK=16,32,64 HRT_REFILL_THRESHOLD=8,16,24
# Prints one line per combination: "<knobs> : Mrays/s ms_per_step roofline.frac".  The sweeps behind the defaults are
# kept under profiles/r01_sweep_*.txt.
OUT=$1; shift
ARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
: > "$OUT"
combos=("")
for spec in "$@"; do
  var=${spec%%=*}; IFS=, read -ra vals <<< "${spec#*=}"
  next=()
  for c in "${combos[@]}"; do for v in "${vals[@]}"; do next+=("$c $var=$v"); done; done
  combos=("${next[@]}")
done
for c in "${combos[@]}"; do
  r=$(env $c python3 bench.py "${ARGS[@]}" --no-cpu-baseline --no-alt-builder 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], 'nodes/ray', d['roofline']['nodes_per_ray'], 'prims/ray', d['roofline']['prims_per_ray'], 'bvh_nodes', d['config']['bvh_nodes'], 'build_s', d['config']['bvh_build_s'])")
  echo "$c : $r" | tee -a "$OUT"
done
