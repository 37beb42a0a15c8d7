#!/bin/bash
# Frames in flight x lanes per frame x hardware queues per process (GPU_MAX_HW_QUEUES, HIP's default is 4): whole-frame
# throughput of the C2 workload.  usage (GPU box, repository root): bash tools/knob_queues.sh [workload]
WL=${1:-c2}
for q in 4 8 16; do
  for fif in 2 3 4; do
    for lanes in 1 2; do
      line=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python bench.py --workload $WL --steps 24 --warmup 4 --no-cpu-baseline --no-brute --frames-in-flight $fif --lanes-per-frame $lanes 2>/dev/null | grep '^{"metric' ) || { echo "queues=$q fif=$fif lanes=$lanes failed"; exit 1; }
      echo "queues=$q frames_in_flight=$fif lanes=$lanes $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step=%.3f Mrays/s=%.0f" % (d["ms_per_step"], d["value"]))')"
    done
  done
done
