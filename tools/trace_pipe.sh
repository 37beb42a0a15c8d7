#!/bin/bash
# Kernel timeline of shares with frames in flight: bash tools/trace_pipe.sh c2 8 4 [queues]   (workload, G, frames in flight, GPU_MAX_HW_QUEUES)
set -e
WL=${1:-c2}; G=${2:-8}; K=${3:-4}; Q=${4:-4}
OUT=gpurun_out/pipe_${WL}_${G}_${K}_q$Q
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT" && mkdir -p "$OUT"
export GPU_MAX_HW_QUEUES=$Q
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python tools/pipeline_run.py "$WL" "$G" "$K" 60 > "$OUT/run.log" 2>&1
f=$(find "$OUT" -name '*kernel_trace.csv' | head -1)
python tools/trace_window.py "$f" 6000 1500 > "$OUT/timeline.txt"
head -90 "$OUT/timeline.txt"
