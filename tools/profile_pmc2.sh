#!/bin/bash
# Second PMC battery: L2 / fabric / busy counters for the traverse kernel.
set -u
OUT=${1:-gpurun_out/pmc3}; shift || true
ARGS=${*:-"--steps 1 --warmup 0 --spp 2 --no-cpu-baseline"}
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {
  name=$1; shift
  echo "pass $name"
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
run tcc1 TCC_BUSY TCC_CYCLE TCC_TAG_STALL TCC_EA0_RDREQ_LEVEL
run tcc2 TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B
run tcc3 TCC_EA0_RDREQ_DRAM TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_GMI_CREDIT_STALL TCC_EA0_RDREQ_IO_CREDIT_STALL
run tcc4 TCC_REQ TCC_READ TCC_STREAMING_REQ TCC_NC_REQ
run grbm1 GRBM_TA_BUSY GRBM_TC_BUSY
run grbm2 GRBM_EA_BUSY GRBM_GUI_ACTIVE
run sq3 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
run sq4 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY
