#!/bin/bash
# Per-launch timeline (with overlaps) of the last frame of one rank's share: bash tools/trace_share.sh c3 8 (on the GPU box).
set -e
WL=${1:-c3}
G=${2:-8}
OUT=gpurun_out/share_${WL}_$G
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT" && mkdir -p "$OUT"
TILE_STEP=$G timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python tools/frame_times.py "$WL" 4 > "$OUT/run.log" 2>&1
f=$(find "$OUT" -name '*kernel_trace.csv' | head -1)
python tools/trace_lanes.py "$f" > "$OUT/timeline.txt"
tail -60 "$OUT/timeline.txt"
