#!/bin/bash
# How busy the vector-memory path (texture addresser TA, vector L1 TCP) is per traversal kernel, frames one at a time:
#   bash tools/pmc_memunit.sh [workload]      (GPU box, repository root; one rocprofv3 --pmc pass, counters only)
# Prints per kernel the fraction of GPU cycles its TAs were busy (TA_TA_BUSY_sum over the CUs' TAs), the TCP's cycles stalled
# on pending misses and on the TA's data path, and vector-L1 accesses and L2 read requests per wave.
WL=${1:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_memunit_$WL
mkdir -p "$OUT"
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d "$OUT/pmc" -- python bench.py --workload $WL --steps 1 --warmup 0 --frames-in-flight 1 --lanes-per-frame 1 --no-cpu-baseline --no-brute --no-extra-legs > "$OUT/pmc.log" 2>&1 || { echo "pmc pass failed"; tail -5 "$OUT/pmc.log"; exit 1; }
python - "$OUT/pmc" <<'PY'
import collections, csv, glob, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rt::k_bvh_" in k and ("closest" in k or "shadow" in k or "tail" in k):
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(tot.items()):
    gui = c["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
    cus = 256.0
    print("%s ta_busy=%.3f (TA_BUSY_avr/gui=%.3f) tcp_pending_stall=%.3f tcp_ta_data_stall=%.3f l1_accesses_per_wave=%.0f l2_read_req_per_wave=%.0f gpu_cycles=%.4g" % (
        k, c["TA_TA_BUSY_sum"] / (cus * gui), c["TA_BUSY_avr"] / gui if gui else 0.0, c["TCP_PENDING_STALL_CYCLES_sum"] / (cus * gui),
        c["TCP_TCP_TA_DATA_STALL_CYCLES_sum"] / (cus * gui), c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["SQ_WAVES"], c["TCP_TCC_READ_REQ_sum"] / c["SQ_WAVES"], gui))
PY
