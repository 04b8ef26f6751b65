"""Aggregates rocprofv3 --pmc counter CSVs (one row per dispatch and counter) per kernel name."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
if len(sys.argv) > 2 and sys.argv[2] == "sq":
    # one pass with several SQ / GRBM counters: per kernel the average per dispatch, plus the derived MFMA pipe
    # occupancy  SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs)
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(out, "pmc_sq", "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            a = agg[row["Kernel_Name"]][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    res = {}
    for k, cs in agg.items():
        e = {c: v[1] / max(v[0], 1) for c, v in cs.items()}
        e["dispatches"] = max(v[0] for v in cs.values())
        if e.get("GRBM_GUI_ACTIVE"):
            e["mfma_pipe_occupancy"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * e["GRBM_GUI_ACTIVE"] / 8)
        res[k] = e
    print(json.dumps(res, indent=1))
    sys.exit(0)
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
