#!/bin/bash
# Per-launch timeline of the last frame of a workload: bash tools/trace_frames.sh c2 (on the GPU box).
set -e
WL=${1:-c2}
OUT=gpurun_out/trace_$WL
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python tools/frame_times.py "$WL" 3 > "$OUT/run.log" 2>&1
f=$(find "$OUT" -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
last = max(i for i, n in enumerate(names) if 'k_raygen' in n or 'k_primary' in n)
prev = None
for r in rows[last:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{r['Kernel_Name'][:34]:34s} grid {r['Grid_Size_X']:>9s} dur_us {(e - s) / 1e3:8.1f} gap_us {gap:7.1f}")
    prev = e
PY
