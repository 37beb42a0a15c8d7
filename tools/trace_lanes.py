"""Timeline of the last frame in a rocprofv3 kernel trace, with overlaps: python tools/trace_lanes.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
starts = [i for i, n in enumerate(names) if 'k_raygen' in n or 'k_bvh_closest_pk' in n or 'k_primary' in n]
# the last frame begins at the first camera-ray kernel of the last group of two (one per lane)
first = starts[-2] if len(starts) >= 2 and int(rows[starts[-1]]['Start_Timestamp']) - int(rows[starts[-2]]['Start_Timestamp']) < 2_000_000 else starts[-1]
t0 = int(rows[first]['Start_Timestamp'])
busy = []
for r in rows[first:]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    busy.append((s, e))
    print(f"{r['Kernel_Name'][9:40]:32s} q{r.get('Queue_Id', '?'):>3s} grid {r['Grid_Size_X']:>9s} start {s / 1e3:8.1f} end {e / 1e3:8.1f} dur {(e - s) / 1e3:7.1f}")
end = max(e for _, e in busy)
# time with 0, 1, 2+ kernels in flight
ev = sorted([(s, 1) for s, _ in busy] + [(e, -1) for _, e in busy])
depth, last, acc = 0, 0, {}
for t, d in ev:
    acc[min(depth, 3)] = acc.get(min(depth, 3), 0) + (t - last)
    depth += d; last = t
print("frame %.1f us; in flight 0/1/2/3+: " % (end / 1e3) + " / ".join("%.0f" % (acc.get(k, 0) / 1e3) for k in range(4)))
