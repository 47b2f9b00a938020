"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, counters summed over dispatches / dispatch count."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:44]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            if "at::" in k: continue
            n = len(disp[k])
            print(d, k, "x%d" % n, {a: round(b / n / 1e6, 3) for a, b in sorted(v.items())})
