#!/bin/bash
# PMC passes on the memory pipeline (address/texture units, L1) of the default bench command.
# usage: bash tools/pmc_mem.sh <tag>    (run from the repository root on the GPU box)
set -e
TAG=${1:-mem}
OUT=gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-brute"
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/p1" -- $B > "$OUT/p1.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum --output-format csv -d "$OUT/p2" -- $B > "$OUT/p2.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TAGRAM0_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --output-format csv -d "$OUT/p3" -- $B > "$OUT/p3.log" 2>&1
echo pmc_done
