"""Sum the counters of rocprofv3 --pmc csv output per kernel: python tools/pmc_table.py <dir> [kernel substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if pat and pat not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in sorted(tot):
    print(k, "dispatches", len(cnt[k]))
    for c in sorted(tot[k]): print("   %-44s %.4g" % (c, tot[k][c]))
