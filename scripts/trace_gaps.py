#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace csv: per queue (= stream) the gaps between a kernel's end and the next
kernel's start, and over all queues the time nothing runs.   usage: trace_gaps.py <dir or kernel_trace.csv> [fraction of the kernels to analyse, from the end; default 0.5]
-> how much of a step is launch gap (what a hipGraph of the forward could remove) against kernel time."""
import csv
import glob
import os
import sys


def main():
    p = sys.argv[1]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5   # analyse the last `frac` of the kernels (the warm-up steps come first)
    if os.path.isdir(p):
        p = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [r for r in csv.DictReader(open(p)) if "asep::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[int(len(rows) * (1.0 - frac)):]
    t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    # union of busy intervals over all queues
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
    union, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    byq = {}
    for r in rows:
        byq.setdefault(r["Queue_Id"], []).append(r)
    print(f"{len(rows)} kernels over {(t1 - t0) / 1e6:.2f} ms: summed kernel time {busy / 1e6:.2f} ms, chip busy (union) {union / 1e6:.2f} ms "
          f"= {union / (t1 - t0):.3f} of the span; idle {(t1 - t0 - union) / 1e6:.2f} ms")
    # idle intervals of the whole chip (no kernel of any queue running), by length: launch gaps are a few microseconds, step boundaries milliseconds
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
    idle, cur_e = [], ev[0][1]
    for s_, e_ in ev[1:]:
        if s_ > cur_e:
            idle.append(s_ - cur_e)
        cur_e = max(cur_e, e_)
    for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 20e3), (20e3, 200e3), (200e3, 1e12)):
        sel = [g for g in idle if lo <= g < hi]
        print(f"  chip-idle intervals {lo / 1e3:g} .. {hi / 1e3:g} us: {len(sel)}, total {sum(sel) / 1e6:.3f} ms")
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        print(f"  queue {q}: {len(rs)} kernels, {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e6:.2f} ms of kernel time")


if __name__ == "__main__":
    main()
