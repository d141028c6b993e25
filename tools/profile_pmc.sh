#!/bin/bash
# PMC passes over the bench (run on the GPU box through gpurun).  One rocprofv3 run per counter
# group (SQ has 8 slots, TCC 4; FETCH_SIZE / WRITE_SIZE need their own passes), --kernel-trace only.
# Usage: tools/profile_pmc.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${*:-"--steps 1 --warmup 0 --spp 2 --no-cpu-baseline"}
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR
run tcp TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TOTAL_READ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES
run ta TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_READ_WAVEFRONTS TA_TOTAL_WAVEFRONTS
run tcc TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
