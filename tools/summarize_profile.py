#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag> (written by tools/profile_round.sh) into profiles/<tag>_*:
kernel stats CSVs (frames in flight; frames one at a time; the default command), every pass's bench line, and per-kernel
PMC totals with derived metrics (VALU issue fraction, live lanes, HBM traffic with the gfx950 FETCH_SIZE correction, L2
hit rate).  Per-frame figures divide by the number of frames the pass itself reports (`frames_rendered` of its bench
line), never by a constant; the clock the counters imply (GRBM_GUI_ACTIVE over the serial trace's kernel time) must
come out at or below the part's 2.4 GHz, or the normalisation is wrong and the script fails."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

MAX_CLOCK_GHZ = 2.6           # a 2.4 GHz part; GUI cycles of a dispatch over its own duration read up to 5 % high on dispatches shorter than 0.3 ms


def newest(pattern):
    """One file per pass directory: gpurun merges each run's files into the local copy, so earlier runs' files stay."""
    by_dir = {}
    for f in glob.glob(pattern):
        top = f.split("/")[2]
        if top not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[top]):
            by_dir[top] = f
    return list(by_dir.values())


def bench_line(log):
    """The JSON line bench.py printed in a pass, or None."""
    if not os.path.exists(log):
        return None
    lines = [l for l in open(log, errors="replace") if l.startswith('{"metric"')]
    return json.loads(lines[-1]) if lines else None


def kname(row):
    return row["Kernel_Name"].split("(")[0].replace("void ", "")


def trace_of(src, which):
    """-> (total ms, calls) per kernel of a kernel-trace pass"""
    dur = collections.defaultdict(float); calls = collections.defaultdict(int)
    for f in newest(f"{src}/{which}/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k = kname(r)
            if "rt::k_" in k:
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; calls[k] += 1
    return dur, calls


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03_bvh"
    src = f"gpurun_out/prof_{tag}"
    algo_key = "scan" if ("brute" in tag or "scan" in tag) else "bvh"
    os.makedirs("profiles", exist_ok=True)
    for which, name in (("trace", "kernel_stats"), ("trace_serial", "serial_kernel_stats"), ("trace_default", "default_cmd_kernel_stats")):
        for f in newest(f"{src}/{which}/*/*kernel_stats.csv"):
            shutil.copy(f, f"profiles/{tag}_{name}.csv")
    lines = {}
    for which, name in (("trace", "trace_cmd_bench"), ("trace_serial", "serial_cmd_bench"), ("trace_default", "default_cmd_bench")):
        lines[which] = bench_line(f"{src}/{which}.log")
        if lines[which]:
            json.dump(lines[which], open(f"profiles/{tag}_{name}.json", "w"))

    def frames_of(which):
        b = bench_line(f"{src}/{which}.log")
        if not b or "frames_rendered" not in b:
            raise SystemExit(f"{src}/{which}.log holds no bench line with frames_rendered: cannot normalise per frame")
        return float(b["frames_rendered"][algo_key])

    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    per_launch = collections.defaultdict(lambda: collections.defaultdict(float))
    pass_frames = {}
    gui_ns = collections.defaultdict(int)
    for f in newest(f"{src}/pmc_*/*/*counter_collection.csv"):
        which = f.split("/")[2]
        pass_frames[which] = frames_of(which)
        raw = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(lambda: collections.defaultdict(int))
        for r in csv.DictReader(open(f)):
            k = kname(r)
            if "rt::k_" not in k:
                continue
            raw[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]] += 1                          # one row per dispatch and counter
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":               # the same dispatches' own durations, for the clock
                gui_ns[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for k in raw:
            for name, v in raw[k].items():
                tot[k][name] += v / pass_frames[which]               # per frame of the pass that collected the counter
                per_launch[k][name] += v / disp[k][name]             # per launch of that pass
    # kernel durations: frames one at a time (what a launch takes alone) and frames in flight
    ser_frames, ovl_frames = frames_of("trace_serial"), frames_of("trace")
    ser_dur, ser_calls = trace_of(src, "trace_serial")
    ovl_dur, ovl_calls = trace_of(src, "trace")
    out = {}
    N_SIMD = 1024
    for k, c in tot.items():
        # kernels of the other algorithm ran for other frame counts (a hierarchy run takes its reference test counts from one
        # scan frame): they are not this summary's subject
        if ("k_bvh_" in k) != (algo_key == "bvh"):
            continue
        d = dict(c)                                          # every counter: per frame
        ms_frame = ser_dur[k] / ser_frames if k in ser_dur else None
        gui = d.get("GRBM_GUI_ACTIVE", 0) / 8.0              # summed over the 8 XCDs
        m = {"counters_per_frame": d, "frames_in_pass": pass_frames,
             "ms_per_frame_serial": ms_frame, "launches_per_frame": ser_calls[k] / ser_frames if k in ser_calls else None,
             "avg_launch_ms_serial": ser_dur[k] / ser_calls[k] if ser_calls.get(k) else None,
             "ms_per_frame_in_flight": ovl_dur[k] / ovl_frames if k in ovl_dur else None,
             "avg_launch_ms_in_flight": ovl_dur[k] / ovl_calls[k] if ovl_calls.get(k) else None}
        if gui:
            m["per_frame_kernel"] = bool(m["launches_per_frame"] and m["launches_per_frame"] >= 1.0)   # else: once per scene (hierarchy build)
            if ms_frame and m["per_frame_kernel"]:
                # GUI cycles per frame of the counter pass over (a) the durations of the same dispatches in that pass: the
                # clock the chip held; (b) the un-profiled serial durations: the same, times the counter pass's slow-down.
                # A frame miscount shows in (b) as a factor, not as per cent.
                m["clock_ghz"] = d["GRBM_GUI_ACTIVE"] * pass_frames["pmc_sq2"] / 8.0 / gui_ns[k] if gui_ns.get(k) else None
                m["clock_ghz_against_unprofiled_durations"] = gui / (ms_frame * 1e6)
                if (m["clock_ghz"] or 0) > MAX_CLOCK_GHZ or m["clock_ghz_against_unprofiled_durations"] > 1.25 * MAX_CLOCK_GHZ:
                    raise SystemExit(f"{k}: the counters imply a clock of {m['clock_ghz']} / {m['clock_ghz_against_unprofiled_durations']:.2f} GHz: frames are miscounted")
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
                m["live_lane_fraction"] = d.get("SQ_THREAD_CYCLES_VALU", 0) / (d["SQ_ACTIVE_INST_VALU"] * 64.0)
            m["occupancy_waves_per_simd"] = d.get("SQ_WAVE_CYCLES", 0) * 4.0 / (gui * N_SIMD)
        if d.get("SQ_WAVE_CYCLES"):
            m["wait_fraction_of_wave_cycles"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
        if d.get("SQ_WAVES"):
            m["valu_insts_per_wave"] = d.get("SQ_INSTS_VALU", 0) / d["SQ_WAVES"]
        if "FETCH_SIZE" in d:
            # FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE reads half of a wide coalesced stream on gfx950 (x2 upper bound)
            m["hbm_read_bytes_low"] = d["FETCH_SIZE"] * 1024.0
            m["hbm_read_bytes_corrected"] = d["FETCH_SIZE"] * 2048.0
        if "WRITE_SIZE" in d:
            m["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024.0
        pl = per_launch[k]
        if "FETCH_SIZE" in pl and "WRITE_SIZE" in pl:
            m["hbm_bytes_per_launch"] = pl["FETCH_SIZE"] * 2048.0 + pl["WRITE_SIZE"] * 1024.0        # per launch of the counters' own passes
        if d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0) > 0:
            m["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
            m["l2_miss_bytes_per_launch"] = pl["TCC_MISS_sum"] * 128.0      # 128-byte lines: cross-check of hbm_bytes_per_launch
        out[k] = m
    json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
    for k, m in sorted(out.items()):
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in m.items() if a not in ("counters_per_frame", "frames_in_pass")})


if __name__ == "__main__":
    main()
