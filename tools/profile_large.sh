#!/bin/bash
# Out-of-cache datapoints of the path kernel (profiles/rNN_large_scenes.txt): for each scene size a plain run (build time, Mrays/s,
# nodes / primitives per ray) and rocprofv3 --pmc passes for the memory side (separate passes, --kernel-trace only; the program
# directly after `--`).   Usage: tools/profile_large.sh r03 "1000000 8000000 32000000"
set -u
TAG=${1:-r03}; SIZES=${2:-"1000000 8000000 32000000"}
OUT=gpurun_out/large; mkdir -p $OUT; export TMPDIR=/tmp
for N in $SIZES; do
  echo "== $N triangles =="
  timeout -k 10 900 python3 tools/large_scene_bench.py --tris $N > $OUT/run_$N.json 2> $OUT/run_$N.err || { echo "run $N failed"; tail -3 $OUT/run_$N.err; exit 1; }
  tail -1 $OUT/run_$N.json
  pmc() { name=$1; shift; timeout -k 10 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_${N}_$name -- python3 tools/large_scene_bench.py --tris $N --steps 1 --no-count > $OUT/pmc_${N}_$name.log 2>&1 || echo "pass $name ($N) failed"; }
  pmc tcc TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ
  pmc rd TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B
  pmc wr WRITE_SIZE
  pmc sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD
  pmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_WR
  pmc grbm GRBM_GUI_ACTIVE
done
python3 - "$TAG" "$SIZES" <<'PY'
import csv, glob, json, collections, sys
tag, sizes = sys.argv[1], sys.argv[2].split()
lines = []
for n in sizes:
    run = json.loads(open(f"gpurun_out/large/run_{n}.json").read().strip().splitlines()[-1])
    acc = collections.defaultdict(list)
    for path in glob.glob(f"gpurun_out/large/pmc_{n}_*/*/*counter_collection.csv"):
        rows = [row for row in csv.DictReader(open(path)) if "k_fused<" in row["Kernel_Name"]]
        # the LAST launch of the process is the timed 16-spp step (the first is the 2-spp warm-up)
        last = max((int(row["Dispatch_Id"]) for row in rows), default=None)
        for row in rows:
            if int(row["Dispatch_Id"]) == last:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    reads = 32 * m.get("TCC_EA0_RDREQ_32B", 0) + 64 * m.get("TCC_EA0_RDREQ_64B", 0) + 128 * m.get("TCC_EA0_RDREQ_128B", 0)
    fabric = reads + 1024 * m.get("WRITE_SIZE", 0)
    hit = m["TCC_HIT"] / (m["TCC_HIT"] + m["TCC_MISS"]) if m.get("TCC_HIT") else None
    kms = run["kernel_ms_per_step"]
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    n_inst = sum(m.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"))
    run.update({"fabric_bytes_per_launch": fabric, "tcc_hit_rate": hit, "measured_frac_of_8TBs": fabric / (kms * 1e-3) / 8e12 if kms else None,
                "fabric_over_algorithmic": fabric / run["algorithmic_bytes_per_launch"] if run.get("algorithmic_bytes_per_launch") else None,
                "lanes_active_frac": m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"]) if m.get("SQ_ACTIVE_INST_VALU") else None,
                "simd_cycles_per_instruction": cyc * 1024.0 / n_inst if n_inst and cyc else None,
                "wave_time_waiting_on_memory": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAIT_ANY") and m.get("SQ_WAVE_CYCLES") else None,
                "wave_time_issuing": m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_ACTIVE_INST_ANY") and m.get("SQ_WAVE_CYCLES") else None})
    lines.append(json.dumps(run))
open(f"gpurun_out/large/{tag}_large_scenes.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
