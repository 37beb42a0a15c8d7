"""Steady-state window of a rocprofv3 kernel trace of tools/pipeline_run.py: launches with their queue, how many kernels
run at once, busy time per queue.  python tools/trace_window.py <kernel_trace.csv> [window_us] [print_us]"""
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rt::k_" in r["Kernel_Name"]]
win = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 6e6
show = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 1.5e6
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = max(int(r["End_Timestamp"]) for r in rows)
lo = t_end - 1e6 - win                      # leave out the drain at the end
hi = t_end - 1e6
sel = [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]
frames = sum(1 for r in sel if "closest_pk" in r["Kernel_Name"])
print(f"window {win / 1e3:.0f} us: {len(sel)} launches, {frames} frames begun -> {win / 1e3 / max(frames, 1):.1f} us per frame")
ev = []
perq = collections.defaultdict(float); perk = collections.defaultdict(float)
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev += [(s, 1), (e, -1)]
    perq[r.get("Queue_Id", "?")] += e - s
    perk[r["Kernel_Name"].split("(")[0].replace("void rt::", "")] += e - s
ev.sort()
depth, last, acc = 0, lo, collections.defaultdict(float)
for t, d in ev:
    acc[min(depth, 6)] += t - last; depth += d; last = t
print("time with n kernels running (us): " + ", ".join(f"{k}: {v / 1e3:.0f}" for k, v in sorted(acc.items())))
print("busy per queue (us): " + ", ".join(f"q{k}: {v / 1e3:.0f}" for k, v in sorted(perq.items())))
for k, v in sorted(perk.items(), key=lambda kv: -kv[1]):
    print(f"  {k:44s} {v / 1e3 / max(frames, 1):8.1f} us per frame (sum of durations)")
for r in sel:
    s, e = int(r["Start_Timestamp"]) - lo, int(r["End_Timestamp"]) - lo
    if s > show: break
    print(f"{r['Kernel_Name'].split('(')[0].replace('void rt::', '')[:40]:40s} q{r.get('Queue_Id', '?'):>3s} grid {r['Grid_Size_X']:>9s} start {s / 1e3:8.1f} end {e / 1e3:8.1f} dur {(e - s) / 1e3:7.1f}")
