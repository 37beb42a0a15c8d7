#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box: kernel trace + stats of the bench command (frames in
# flight, as timed), the same with frames one at a time on one stream (launches that do not overlap: the durations
# the roofline fraction is priced on), then PMC passes (each in a run of its own, counters only), then the default
# command as the driver runs it.  Every pass's bench line (with its count of rendered frames) is kept beside the
# pass's output; tools/summarize_profile.py turns the lot into profiles/<tag>_*.
# usage: bash tools/profile_round.sh r03_bvh bvh   (tag, algorithm; run from the repository root on the GPU box)
set -e
TAG=${1:-r03_bvh}
ALGO=${2:-bvh}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
COMMON="--no-cpu-baseline --no-brute --no-extra-legs --algo $ALGO"
SERIAL="--frames-in-flight 1 --lanes-per-frame 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python bench.py --steps 4 --warmup 1 $COMMON > "$OUT/trace.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_serial" -- python bench.py --steps 12 --warmup 1 $SERIAL $COMMON > "$OUT/trace_serial.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d "$OUT/pmc_sq" -- python bench.py --steps 1 --warmup 0 $SERIAL $COMMON > "$OUT/pmc_sq.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_sq2" -- python bench.py --steps 1 --warmup 0 $SERIAL $COMMON > "$OUT/pmc_sq2.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python bench.py --steps 1 --warmup 0 $SERIAL $COMMON > "$OUT/pmc_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python bench.py --steps 1 --warmup 0 $SERIAL $COMMON > "$OUT/pmc_write.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/pmc_l2" -- python bench.py --steps 1 --warmup 0 $SERIAL $COMMON > "$OUT/pmc_l2.log" 2>&1 || echo "l2 pass failed"
echo profile_done
# the default command as the driver runs it (throughput mode, every leg, brute-force leg and CPU baseline included): kernel trace + stats
if [ "$ALGO" = "bvh" ]; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python bench.py > "$OUT/trace_default.log" 2>&1
fi
echo profile_done
