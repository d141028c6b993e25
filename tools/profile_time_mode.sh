#!/bin/bash
# profiles/rNN_time_mode_driver.txt + rNN_time_mode_kernel_stats.csv: the reference's own frame loop (hrt_time_render on its shipped sample, 27 frames
# 1200x800) in the default configuration and with the knobs that used to be faster, the per-step breakdown, a kernel trace of the loop, and the synthetic
# Time-mode loops at 25 / 2000 particles (tools/time_mode_bench.py).  Usage: tools/profile_time_mode.sh r04
set -u
TAG=${1:-r04}; OUT=gpurun_out/final; mkdir -p $OUT; export TMPDIR=/tmp
B=$PWD/nvidia-optix-ray-tracer_amd/lib/hrt_time_render; ARGS="tests/golden/files/config.json tests/golden/files -1 /tmp/o.ppm"
{
echo "# hrt_time_render $ARGS (the reference's shipped sample: 1 ground sphere, 8 STL shapes, 3 files x 9 frames of 25 particles, 1200x800, 1 spp), build $(cat nvidia-optix-ray-tracer_amd/lib/BUILD_ID)"
echo "== default configuration (three runs)"; for i in 1 2 3; do $B $ARGS 2>&1 | tail -1; done
echo "== HRT_TIME_RENDER_BREAKDOWN=1 (every step bracketed by device synchronisations)"; HRT_TIME_RENDER_BREAKDOWN=1 $B $ARGS 2>&1 | tail -3
echo "== HRT_BUILD=host"; for i in 1 2; do HRT_BUILD=host $B $ARGS 2>&1 | tail -1; done
echo "== HRT_TLAS_INSTANCED=-1 (rebuilds are merged device builds)"; for i in 1 2; do HRT_TLAS_INSTANCED=-1 $B $ARGS 2>&1 | tail -1; done
echo "== HRT_REFIT_MOVED_FAR=0 (round 3's policy: refit the identity-built tree, check, rebuild)"; for i in 1 2; do HRT_REFIT_MOVED_FAR=0 $B $ARGS 2>&1 | tail -1; done
echo "== HRT_BVH_CPRIM=0.45 (round 3's leaves)"; for i in 1 2; do HRT_BVH_CPRIM=0.45 $B $ARGS 2>&1 | tail -1; done
echo "== HRT_TIME_RENDER_SYNC_UPDATE=1 (synchronous updates)"; HRT_TIME_RENDER_SYNC_UPDATE=1 $B $ARGS 2>&1 | tail -1
} > $OUT/${TAG}_time_mode_driver.txt 2>&1
( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/tm_trace -- $B $OLDPWD/tests/golden/files/config.json $OLDPWD/tests/golden/files -1 /tmp/o.ppm > $OLDPWD/$OUT/tm_trace.log 2>&1 )
cp $OUT/tm_trace/*/*kernel_stats.csv $OUT/${TAG}_time_mode_kernel_stats.csv 2>/dev/null
python3 - "$OUT" "$TAG" <<'PY' >> $OUT/${TAG}_time_mode_driver.txt
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
paths = glob.glob(f"{out}/tm_trace/*/*kernel_trace.csv")
if paths:
    k = sorted(csv.DictReader(open(paths[0])), key=lambda r: int(r["Start_Timestamp"]))
    fused = [i for i, r in enumerate(k) if "k_fused" in r["Kernel_Name"]]
    i0, i1 = fused[-5], fused[-4]
    t0 = int(k[i0]["Start_Timestamp"]); prev = None
    print("== one ordinary frame of the loop on the device (rocprofv3 --kernel-trace; start, gap to the previous end, duration in microseconds)")
    for r in k[i0:i1 + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  {(s - t0) / 1e3:8.1f}  gap {((s - prev) / 1e3 if prev else 0):6.1f}  {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:80]}")
        prev = e
PY
{ echo "== tools/time_mode_bench.py (synthetic Time-mode loops, 1200x800, 200 frames)"; timeout -k 10 300 python3 tools/time_mode_bench.py 2>&1 | grep -v amdgpu.ids; } >> $OUT/${TAG}_time_mode_driver.txt
cat $OUT/${TAG}_time_mode_driver.txt
