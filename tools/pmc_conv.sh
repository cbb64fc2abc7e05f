#!/bin/bash
# PMC passes over the dominant convolution kernel (tests/probes/conv_pmc_target.py), one rocprofv3 --pmc run per counter
# group with --kernel-trace only (no other trace domain), as MI355X_MICROARCH.md prescribes.  GPU box, repo root:
#     bash tools/pmc_conv.sh <tag> [conv_pmc_target.py arguments ...]
# -> gpurun_out/pmc_conv/<tag>/<group>/ ... and gpurun_out/pmc_conv/<tag>.json (tools/pmc_conv_table.py)
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_conv/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
declare -A G
G[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_LDS"
G[sq2]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM"
G[sq3]="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
G[tcp1]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
G[tcp2]="TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum"
G[tcc1]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
G[tcc2]="TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_WRREQ_sum"
# (TA_* / TD_* groups of four do not fit one pass on gfx950 - "Request exceeds the capabilities of the hardware" - and rocprofv3
#  then waits forever for a child that is gone: every pass runs under `timeout`)
G[grbm]="GRBM_GUI_ACTIVE GRBM_COUNT"
for g in ${PMC_GROUPS:-sq1 sq2 sq3 tcp1 tcp2 tcc1 tcc2 grbm}; do
  rm -rf "$OUT/$g"
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc ${G[$g]} -d "$OUT/$g" -o run --output-format csv -- \
      python "$REPO/tests/probes/conv_pmc_target.py" "$@" > "$OUT/$g.log" 2>&1
  echo "$g rc=$?"
  find "$OUT/$g" -name "*agent_info.csv" -delete
done
python "$REPO/tools/pmc_conv_table.py" "$OUT" > "$OUT.json"
find "$OUT" -name "*.csv" -size +2M -delete
cat "$OUT.json"
