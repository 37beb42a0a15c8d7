#!/bin/bash
# n short bench.py runs of a workload, one line each: bash tools/bench_reps.sh c2 3 [extra bench.py args]
WL=${1:-c2}; N=${2:-3}; shift; shift
for rep in $(seq $N); do
  line=$(timeout -k 10 120 python bench.py --workload $WL --steps 40 --warmup 4 --no-cpu-baseline --no-brute "$@" 2>/dev/null | grep '^{"metric') || { echo "run failed"; exit 1; }
  echo "$WL rep=$rep $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms_per_step=%.3f single=%.3f Mrays/s=%.0f serial_ms=%s walk=%s ovf=%s" % (d["ms_per_step"], d.get("single_frame_ms") or 0.0, d["value"], {k.split()[0]: round(v,3) for k,v in r["kernel_ms_per_step"].items()}, {k.split()[0]: (round(v["nodes"],2), round(v["exact_tests"],2)) for k,v in (r.get("walk_per_ray") or {}).items()}, r.get("stack_overflows_per_step")))')"
done
