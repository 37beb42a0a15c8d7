#!/bin/bash
# Per-kernel time of a few frames of one workload: bash tools/trace_frames.sh c2 (on the GPU box).
set -e
WL=${1:-c2}
OUT=gpurun_out/trace_$WL
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python tools/frame_times.py "$WL" 3 > "$OUT/run.log" 2>&1
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:10.1f} {r['Percentage']}%")
PY
