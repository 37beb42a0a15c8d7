#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag> (written by tools/profile_round.sh) into profiles/<tag>_*:
kernel stats CSV, per-kernel PMC totals and derived metrics (VALU utilisation, HBM traffic with
the gfx950 FETCH_SIZE correction, L2 hit rate)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

def newest(pattern):
    """One file per pass directory: gpurun merges each run's files into the local copy, so earlier runs' files stay."""
    by_dir = {}
    for f in glob.glob(pattern):
        top = f.split("/")[2]
        if top not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[top]):
            by_dir[top] = f
    return list(by_dir.values())


tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
for f in newest(f"{src}/trace/*/*kernel_stats.csv"):
    shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
# the default command (python bench.py, as the driver runs it): its kernel stats and its JSON line
for f in newest(f"{src}/trace_default/*/*kernel_stats.csv"):
    shutil.copy(f, f"profiles/{tag}_default_cmd_kernel_stats.csv")
for log, name in ((f"{src}/trace_default.log", f"profiles/{tag}_default_cmd_bench.json"), (f"{src}/trace.log", f"profiles/{tag}_trace_cmd_bench.json")):
    if os.path.exists(log):
        lines = [l for l in open(log, errors="replace") if l.startswith('{"metric"')]
        if lines:
            open(name, "w").write(lines[-1])
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(int)
for f in newest(f"{src}/pmc_*/*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rt::k_" not in k:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if "pmc_sq/" in f.replace("\\", "/") and r["Counter_Name"] == "SQ_WAVES":
            disp[k] += 1
# kernel durations from the trace (same command line: 1 warm-up + 1 probe + 2 timed frames)
dur = collections.defaultdict(float); calls = collections.defaultdict(int)
for f in newest(f"{src}/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rt::k_" in k:
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; calls[k] += 1
out = {}
N_SIMD = 1024
# bench.py renders warm-up frames, one probe frame (every class timed) and the timed steps with the chosen algorithm:
# --steps 2 --warmup 1 in the trace run, --steps 1 --warmup 0 in the PMC runs
TRACE_FRAMES, PMC_FRAMES = 4.0, 2.0
for k, c in tot.items():
    d = dict(c)
    ms_frame = dur[k] / TRACE_FRAMES if k in dur else None
    gui = d.get("GRBM_GUI_ACTIVE", 0) / 8.0                 # summed over the 8 XCDs
    m = {"counters": d, "counters_cover_frames": PMC_FRAMES, "ms_per_frame_from_trace": ms_frame,
         "launches_per_frame": calls[k] / TRACE_FRAMES if k in calls else None}
    if gui:
        m["clock_ghz"] = gui / (ms_frame * PMC_FRAMES * 1e6) if ms_frame else None
        # the SIMDs are 32 lanes wide: with two or more waves resident a SIMD issues one wave64 VALU instruction per
        # 2 cycles (the 157.3 TFLOP/s FP32 peak is that rate), so the fraction of the issue peak is instructions x 2 /
        # SIMD-cycles.  SQ_ACTIVE_INST_VALU (quad-cycles a wave spends in VALU instructions, 4 cycles each for a wave
        # alone) over the same cycles can exceed 1 when waves overlap and is kept under its own name.
        m["valu_issue_fraction"] = d.get("SQ_INSTS_VALU", 0) * 2.0 / (gui * N_SIMD)
        m["valu_wave_busy"] = d.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / (gui * N_SIMD)
        # rocprofiler-sdk's own derived metrics for gfx950 (share/rocprofiler-sdk/counter_defs.yaml):
        #   VALUBusy = 100 * sum(SQ_ACTIVE_INST_VALU) / CU_NUM / max(GRBM_GUI_ACTIVE)   (= valu_wave_busy in percent; it can
        #   exceed 100 because two waves of a SIMD overlap their VALU instructions)
        #   VALUUtilization = 100 * sum(SQ_THREAD_CYCLES_VALU) / (sum(SQ_ACTIVE_INST_VALU) * 64)   (live lanes)
        m["rocprof_VALUBusy_pct"] = 100.0 * d.get("SQ_ACTIVE_INST_VALU", 0) / 256.0 / gui
        if d.get("SQ_ACTIVE_INST_VALU"):
            m["rocprof_VALUUtilization_pct"] = 100.0 * d.get("SQ_THREAD_CYCLES_VALU", 0) / (d["SQ_ACTIVE_INST_VALU"] * 64.0)
        m["occupancy_waves_per_simd"] = d.get("SQ_WAVE_CYCLES", 0) * 4.0 / (gui * N_SIMD)
    if "FETCH_SIZE" in d:
        # FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE reads half of a wide coalesced stream on gfx950 (x2 upper bound)
        # (bytes per frame)
        m["hbm_read_bytes_low"] = d["FETCH_SIZE"] * 1024.0 / PMC_FRAMES
        m["hbm_read_bytes_corrected"] = d["FETCH_SIZE"] * 2048.0 / PMC_FRAMES
    if "WRITE_SIZE" in d:
        m["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024.0 / PMC_FRAMES
    if d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0) > 0:
        m["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    out[k] = m
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, m in sorted(out.items()):
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in m.items() if a != "counters"})
