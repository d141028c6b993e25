#!/bin/bash
# SQ counter passes only (two rocprofv3 runs of the bench), for comparing kernel variants quickly.
# Usage: tools/profile_pmc_sq.sh <outdir> [bench args...]     (environment knobs such as HRT_FUSED are inherited)
set -u
OUT=${1:-gpurun_out/pmc_sq}; shift || true
ARGS=${*:-"--steps 1 --warmup 0 --spp 8 --no-cpu-baseline"}
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
run sq3 SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_FLAT SQ_INSTS_VSKIPPED SQ_INSTS_VALU_MFMA_I8
run sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run grbm GRBM_GUI_ACTIVE
