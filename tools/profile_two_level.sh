#!/bin/bash
# profiles/rNN_instanced.txt: two-level against flattened trees on the particle clouds (2000 / 10^5 particles of three shared shapes): memory, build and
# update cost, Mrays/s (tools/two_level_bench.py), executed node steps / primitive tests / instances entered per ray (the counters build), and the memory side
# of the path kernel from rocprofv3 --pmc passes (separate passes, --kernel-trace only, the program directly after `--`): TCC hit rate, fabric bytes per ray.
# Usage: tools/profile_two_level.sh r04
set -u
TAG=${1:-r04}; OUT=gpurun_out/two_level; mkdir -p $OUT; export TMPDIR=/tmp
L=$PWD/nvidia-optix-ray-tracer_amd/lib
timeout -k 10 400 python3 tools/two_level_bench.py > $OUT/bench_sync.jsonl 2> $OUT/bench_sync.err
timeout -k 10 400 python3 tools/two_level_bench.py --async-update --spp 1 > $OUT/bench_async.jsonl 2> $OUT/bench_async.err
HRT_LIB=$L/libhrt_stats.so timeout -k 10 400 python3 tools/two_level_bench.py --render-only --spp 4 > $OUT/bench_stats.jsonl 2> $OUT/bench_stats.err
for N in 2000 100000; do for S in flat two; do
  pmc() { name=$1; shift; timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_${N}_${S}_$name -- python3 tools/two_level_bench.py --render-only --spp 4 --particles $N --structures $S > $OUT/pmc_${N}_${S}_$name.log 2>&1 || echo "pass $name ($N $S) failed"; }
  pmc tcc TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ
  pmc rd TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B
done; done
python3 - "$TAG" <<'PY'
import csv, glob, json, collections, sys
tag = sys.argv[1]
def lines(path):
    out = []
    for l in open(path):
        try: out.append(json.loads(l))
        except Exception: pass
    return out
rows = {}
for d in lines("gpurun_out/two_level/bench_sync.jsonl"): rows[(d["instances"] - 1, d["structure"])] = d
for d in lines("gpurun_out/two_level/bench_async.jsonl"): rows[(d["instances"] - 1, d["structure"])]["tlas_update_async_ms"] = d.get("tlas_update_ms")
for d in lines("gpurun_out/two_level/bench_stats.jsonl"): rows[(d["instances"] - 1, d["structure"])]["counters_spp4"] = d["spp4"]
for (n, s), d in rows.items():
    acc = collections.defaultdict(list)
    for path in glob.glob(f"gpurun_out/two_level/pmc_{n}_{s}_*/*/*counter_collection.csv"):
        rr = [r for r in csv.DictReader(open(path)) if "k_fused<" in r["Kernel_Name"]]
        if not rr: continue
        # the 4-spp launches of the timed loop are the largest dispatches: take the last one
        last = max(int(r["Dispatch_Id"]) for r in rr)
        for r in rr:
            if int(r["Dispatch_Id"]) == last: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    rays = d["spp4"]["rays_per_path"] * 1920 * 1080 * 4
    reads = 32 * m.get("TCC_EA0_RDREQ_32B", 0) + 64 * m.get("TCC_EA0_RDREQ_64B", 0) + 128 * m.get("TCC_EA0_RDREQ_128B", 0)
    d["tcc_hit_rate"] = round(m["TCC_HIT"] / (m["TCC_HIT"] + m["TCC_MISS"]), 4) if m.get("TCC_HIT") else None
    d["fabric_read_bytes_per_ray"] = round(reads / rays, 1) if reads else None
    d["fabric_read_GBps"] = round(reads / (d["spp4"]["ms_per_launch"] * 1e-3) / 1e9, 1) if reads else None
out = "\n".join(json.dumps(rows[k]) for k in sorted(rows))
open(f"gpurun_out/two_level/{tag}_instanced.jsonl", "w").write(out + "\n")
print(out)
PY
