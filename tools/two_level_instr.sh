#!/bin/bash
# Dynamic instructions of the path kernel per ray on a particle cloud, flattened against two-level (rocprofv3 --pmc, --kernel-trace only; the last 4-spp launch).
# Usage (GPU box): tools/two_level_instr.sh [particles=2000] > gpurun_out/two_level_instr.txt
set -u
N=${1:-2000}; OUT=gpurun_out/tl_instr; mkdir -p $OUT; export TMPDIR=/tmp
for S in flat two; do
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/$S -- python3 tools/two_level_bench.py --render-only --spp 4 --particles $N --structures $S > $OUT/$S.log 2>&1 || { echo "pass $S failed"; exit 1; }
done
python3 - "$N" <<'PY'
import csv, glob, json, sys, collections
n = sys.argv[1]
for s in ("flat", "two"):
    d = [json.loads(l) for l in open(f"gpurun_out/tl_instr/{s}.log") if l.startswith("{")][-1]
    rays = d["spp4"]["rays_per_path"] * 1920 * 1080 * 4
    acc = collections.defaultdict(float)
    for path in glob.glob(f"gpurun_out/tl_instr/{s}/*/*counter_collection.csv"):
        rr = [r for r in csv.DictReader(open(path)) if "k_fused<" in r["Kernel_Name"]]
        last = max(int(r["Dispatch_Id"]) for r in rr)
        for r in rr:
            if int(r["Dispatch_Id"]) == last: acc[r["Counter_Name"]] += float(r["Counter_Value"])
    tot = acc["SQ_INSTS_VALU"] + acc["SQ_INSTS_SALU"] + acc["SQ_INSTS_LDS"] + acc["SQ_INSTS_VMEM_RD"] + acc["SQ_INSTS_SMEM"]
    print(f"cloud-{n} {s}: {d['spp4']['Mrays_per_s']} Mrays/s; wave instructions per ray: VALU {acc['SQ_INSTS_VALU'] * 64 / rays / 64:.1f}... total {tot / rays:.2f} per ray-lane-64th", flush=True)
    print("   ", {k: round(v / rays, 3) for k, v in acc.items()}, "lanes active", round(acc["SQ_THREAD_CYCLES_VALU"] / (64 * acc["SQ_ACTIVE_INST_VALU"]), 3) if acc["SQ_ACTIVE_INST_VALU"] else None)
PY
