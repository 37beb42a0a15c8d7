#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box: kernel trace + stats of the
# default bench command, then PMC passes (each in a run of its own, counters only).
# usage: bash tools/profile_round.sh r01_bvh bvh   (tag, algorithm; run from the repository root on the GPU box)
set -e
TAG=${1:-r01_bvh}
ALGO=${2:-bvh}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-brute --algo $ALGO > "$OUT/trace.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d "$OUT/pmc_sq" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-brute --algo $ALGO > "$OUT/pmc_sq.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_sq2" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-brute --algo $ALGO > "$OUT/pmc_sq2.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-brute --algo $ALGO > "$OUT/pmc_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-brute --algo $ALGO > "$OUT/pmc_write.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/pmc_l2" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-brute --algo $ALGO > "$OUT/pmc_l2.log" 2>&1 || echo "l2 pass failed"
echo profile_done
# the default command as the driver runs it (throughput mode, brute-force leg and CPU baseline included): kernel trace + stats
if [ "$ALGO" = "bvh" ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python bench.py > "$OUT/trace_default.log" 2>&1
fi
echo profile_done
