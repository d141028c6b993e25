#!/bin/bash
# Memory-side traffic of the dominant kernel (fused path mode of k_traverse), per launch, from PMC counters:
#   reads  = 32*TCC_EA0_RDREQ_32B + 64*TCC_EA0_RDREQ_64B + 128*TCC_EA0_RDREQ_128B   (exact request sizes;
#            FETCH_SIZE tallies every request at 64 B and under-reports this kernel's 128-B requests by 2x)
#   writes = WRITE_SIZE * 1024
# Writes profiles/traverse_traffic.json, which bench.py reports as roofline.traffic.
set -u
OUT=gpurun_out/traffic
ARGS="--steps 1 --warmup 0 --no-cpu-baseline"
mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B --kernel-trace --output-format csv -d $OUT/rd -- python3 bench.py $ARGS > $OUT/rd.log 2>&1
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/wr -- python3 bench.py $ARGS > $OUT/wr.log 2>&1
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fs -- python3 bench.py $ARGS > $OUT/fs.log 2>&1
python3 - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(list)
for path in glob.glob("gpurun_out/traffic/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(path)):
        if "k_traverse<false, false, false, true>" in row["Kernel_Name"]:      # the fused path kernel (COUNT, SPHERES, DMA off; FUSED on)
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
reads = 32 * m.get("TCC_EA0_RDREQ_32B", 0) + 64 * m.get("TCC_EA0_RDREQ_64B", 0) + 128 * m.get("TCC_EA0_RDREQ_128B", 0)
writes = 1024 * m.get("WRITE_SIZE", 0)
out = {"kernel": "k_traverse (dominant instantiation)", "workload": "C4, 1920x1080, 256 spp: one launch of the fused path kernel = one bench step",
       "counters_mean_per_launch": m, "read_bytes_per_launch": reads, "write_bytes_per_launch": writes,
       "fetch_size_kb_uncorrected": m.get("FETCH_SIZE"), "hbm_bytes_per_launch": reads + writes,
       "note": "memory-side (fabric) requests: Infinity-Cache hits are included, so this is an upper bound on HBM bytes"}
json.dump(out, open("gpurun_out/traffic/traverse_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
