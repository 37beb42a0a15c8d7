#!/bin/bash
# A/B of one environment knob on whole frames, with the counters behind it.
# usage (GPU box, repository root): bash tools/ab_env.sh HIPEXEC_BIN_BOUNCE 0 1 [workload]
# Prints ms per frame of alternating bench.py runs (3 per value) and, from one rocprofv3 --pmc pass per value, live lanes per
# issued VALU instruction (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU), resident waves per SIMD and VALU instructions per
# kernel of the traversal kernels.
VAR=$1; A=$2; B=$3; WL=${4:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ab_${VAR}_${WL}
mkdir -p "$OUT"
for rep in 1 2 3; do
  for v in $A $B; do
    line=$(env $VAR=$v timeout -k 10 120 python bench.py --workload $WL --steps 40 --warmup 4 --no-cpu-baseline --no-brute 2>/dev/null | grep '^{"metric') || { echo "$VAR=$v failed"; exit 1; }
    echo "$VAR=$v rep=$rep $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms_per_step=%.3f Mrays/s=%.0f isolated_ms=%s" % (d["ms_per_step"], d["value"], {k.split()[0]: round(v,3) for k,v in r["kernel_ms_per_step"].items()}))')"
  done
done
for v in $A $B; do
  export $VAR=$v
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_$v" -- python bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-brute > "$OUT/pmc_$v.log" 2>&1 || { echo "pmc pass $v failed"; exit 1; }
  python - "$OUT/pmc_$v" "$VAR=$v" <<'PY'
import collections, csv, glob, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rt::k_bvh_" in k and ("closest" in k or "shadow" in k or "tail" in k):
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(tot.items()):
    gui = c["GRBM_GUI_ACTIVE"] / 8.0
    print("%s %s live_lanes=%.3f waves_per_simd=%.2f valu_issue_fraction=%.3f valu_insts=%.4g gpu_cycles=%.4g wait_fraction=%.2f" % (
        sys.argv[2], k, c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), c["SQ_WAVE_CYCLES"] * 4.0 / (gui * 1024.0),
        c["SQ_INSTS_VALU"] * 2.0 / (gui * 1024.0), c["SQ_INSTS_VALU"], gui, c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
done
echo ab_done
