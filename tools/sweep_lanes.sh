#!/bin/bash
# Lane utilisation and instruction mix of the path kernel per knob combination: one rocprofv3 --pmc pass (SQ counters,
# --kernel-trace only) of a short bench run each.
#   tools/sweep_lanes.sh <outfile> [bench args] -- VAR1=a,b VAR2=x,y
# Prints: knobs : kernel ms | lanes-active (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU) | VALU / SALU instructions per launch
OUT=$1; shift
ARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
[ ${#ARGS[@]} -eq 0 ] && ARGS=(--steps 1 --warmup 0 --spp 16 --no-cpu-baseline)
export TMPDIR=/tmp
: > "$OUT"
combos=("")
for spec in "$@"; do
  var=${spec%%=*}; IFS=, read -ra vals <<< "${spec#*=}"
  next=()
  for c in "${combos[@]}"; do for v in "${vals[@]}"; do next+=("$c $var=$v"); done; done
  combos=("${next[@]}")
done
i=0
for c in "${combos[@]}"; do
  d=gpurun_out/lanes_pmc/$i; i=$((i+1)); rm -rf $d; mkdir -p $d
  for kv in $c; do export "$kv"; done
  timeout -k 10 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $d -- python3 bench.py "${ARGS[@]}" > $d/log.txt 2>&1
  r=$(python3 - $d <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = 0; dur = []
for path in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(path)):
        if any(k in row["Kernel_Name"] for k in ("k_fused<", "k_traverse<false, false, false, true>")):
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
for path in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(path)):
        if any(k in row["Kernel_Name"] for k in ("k_fused<", "k_traverse<false, false, false, true>")):
            dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
la = acc["SQ_THREAD_CYCLES_VALU"] / (64 * acc["SQ_ACTIVE_INST_VALU"]) if acc["SQ_ACTIVE_INST_VALU"] else float("nan")
print("%.2f ms | lanes-active %.3f | VALU %.3e SALU %.3e" % (sum(dur), la, acc["SQ_INSTS_VALU"], acc["SQ_INSTS_SALU"]))
PY
)
  echo "$c : $r" | tee -a "$OUT"
done
