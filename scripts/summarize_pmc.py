"""Aggregates rocprofv3 --pmc counter CSVs (one row per dispatch and counter) per kernel name."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
res = {}
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(out, tag, "*", "*counter_collection.csv"))
    agg = defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
    res[counter] = {k: {"dispatches": v[0], "sum": v[1], "avg_per_dispatch": v[1] / max(v[0], 1)} for k, v in agg.items()}
print(json.dumps(res, indent=1))
