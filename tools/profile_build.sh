#!/bin/bash
# tools/profile_build.sh OUTDIR [MODES] -- rocprofv3 kernel stats of tools/build_bench.py (the acceleration-structure build, C4)
OUT=$1; MODES=${2:-split}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $OUT
HRT_BUILD_BENCH_MODES=$MODES timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/build_bench.py C4 4 > $OUT/build_bench.log 2>&1
grep "build:" $OUT/build_bench.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open(sys.argv[1] + "/kernel_stats.txt", "w") as o:
        for r in rows[:30]:
            line = f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:9.3f} ms avg {float(r["AverageNs"])/1e3:9.1f} us'
            print(line); o.write(line + "\n")
PY
