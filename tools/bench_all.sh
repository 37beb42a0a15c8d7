#!/bin/bash
# bench.py lines of every workload for a round's profiles: bash tools/bench_all.sh r03   (GPU box, repository root)
TAG=${1:-r03}
mkdir -p gpurun_out/bench_$TAG
for wl in c1 c3 c4 c5; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-brute 2>/dev/null | grep '^{"metric' > gpurun_out/bench_$TAG/${TAG}_bvh_bench_$wl.json || { echo "$wl failed"; exit 1; }
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['whole_frame']['frac'])" gpurun_out/bench_$TAG/${TAG}_bvh_bench_$wl.json $wl
done
timeout -k 10 600 python bench.py 2>/dev/null | grep '^{"metric' > gpurun_out/bench_$TAG/${TAG}_bench_c2.json || { echo "c2 failed"; exit 1; }
python -c "import json,sys; d=json.load(open(sys.argv[1])); print('c2', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['whole_frame'], d['roofline_brute']['frac'], d['cpu_baseline'])" gpurun_out/bench_$TAG/${TAG}_bench_c2.json
timeout -k 10 300 python bench.py --algo scan --no-cpu-baseline --no-brute 2>/dev/null | grep '^{"metric' > gpurun_out/bench_$TAG/${TAG}_scan_bench_c2.json || { echo "scan failed"; exit 1; }
python -c "import json,sys; d=json.load(open(sys.argv[1])); print('c2 scan', d['ms_per_step'], d['value'])" gpurun_out/bench_$TAG/${TAG}_scan_bench_c2.json
