#!/bin/bash
# The round's evidence for the shipped build, taken on the GPU box in one go (gpurun):
#   profiles/rNN_kernel_stats_c4_256spp.csv     rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 3 --warmup 1`
#   profiles/rNN_pmc_fused_kernel.json          SQ counters of the dominant kernel (separate --pmc passes, --kernel-trace only)
#   profiles/rNN_traverse_traffic.json          memory-side (fabric) bytes per launch: exact request sizes + WRITE_SIZE
#   profiles/rNN_lane_stats.txt                 lane utilisation per phase (instrumented build, make stats)
# Usage: tools/profile_final.sh r02      (writes under gpurun_out/final/, copy what is wanted into profiles/)
set -u
TAG=${1:-r02}
OUT=gpurun_out/final; mkdir -p $OUT; export TMPDIR=/tmp
BUILD_ID=$(cat nvidia-optix-ray-tracer_amd/lib/BUILD_ID 2>/dev/null || echo unknown)
ARGS1="--steps 1 --warmup 0 --no-cpu-baseline --no-alt-builder"
echo "build $BUILD_ID"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-builder > $OUT/kt.log 2>&1
cp $OUT/kt/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_c4_256spp.csv 2>/dev/null
pmc() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 bench.py $ARGS1 > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY
pmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
pmc tcc TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ
pmc rd TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B
pmc wr WRITE_SIZE
pmc fs FETCH_SIZE
python3 - "$TAG" "$BUILD_ID" <<'PY'
import csv, glob, json, collections, sys, time
tag, build = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list); kernel_names = set()
for path in glob.glob("gpurun_out/final/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(path)):
        if "k_fused<" in row["Kernel_Name"] or "k_traverse<false, false, false, true>" in row["Kernel_Name"]:
            kernel_names.add(row["Kernel_Name"])
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
prov = {"build": build, "taken": time.strftime("%Y-%m-%d %H:%M:%S"), "command": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline  (tools/profile_final.sh)",
        "workload": "C4, 1920x1080, 256 spp: one launch of the fused path kernel = one bench step", "kernel": " / ".join(sorted(kernel_names))}
cycles = m.get("GRBM_GUI_ACTIVE", 0) / 8.0                        # the counter is the sum over the 8 XCDs
simds = 256 * 4
valu = {"kernel_cycles": cycles,
        "issue_slot_frac": (m.get("SQ_INSTS_VALU", 0) * 4.0 / (cycles * simds)) if cycles else None,      # one wave64 VALU instruction = 4 cycles of its SIMD's issue port
        "lanes_active_frac": (m.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * m["SQ_ACTIVE_INST_VALU"])) if m.get("SQ_ACTIVE_INST_VALU") else None,
        "wave_time_split": {k: (m.get(c, 0) / m["SQ_WAVE_CYCLES"]) for k, c in (("waiting_on_memory", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY"))} if m.get("SQ_WAVE_CYCLES") else None}
json.dump({"provenance": prov, "counters_mean_per_launch": m, "valu": valu}, open(f"gpurun_out/final/{tag}_pmc_fused_kernel.json", "w"), indent=1)
reads = 32 * m.get("TCC_EA0_RDREQ_32B", 0) + 64 * m.get("TCC_EA0_RDREQ_64B", 0) + 128 * m.get("TCC_EA0_RDREQ_128B", 0)
writes = 1024 * m.get("WRITE_SIZE", 0)
json.dump({"provenance": prov, "fabric_read_bytes_per_launch": reads, "fabric_write_bytes_per_launch": writes, "fabric_bytes_per_launch": reads + writes,
           "fetch_size_kb_uncorrected": m.get("FETCH_SIZE"), "tcc_hit_rate": (m["TCC_HIT"] / (m["TCC_HIT"] + m["TCC_MISS"])) if m.get("TCC_HIT") else None,
           "note": "requests of the L2s to the fabric (exact sizes TCC_EA0_RDREQ_{32,64,128}B; FETCH_SIZE counts this kernel's 128-B requests at 64 B): Infinity-Cache hits are included, so this is an upper bound on HBM bytes"},
          open(f"gpurun_out/final/{tag}_traverse_traffic.json", "w"), indent=1)
print(json.dumps(valu)); print("fabric bytes per launch", reads + writes)
PY
HRT_LIB=$PWD/nvidia-optix-ray-tracer_amd/lib/libhrt_stats.so python3 tools/lane_stats.py > $OUT/${TAG}_lane_stats.txt 2>&1
grep -v amdgpu.ids $OUT/${TAG}_lane_stats.txt
