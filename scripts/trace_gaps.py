#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace csv: per queue (= stream) the gaps between a kernel's end and the next
kernel's start, and over all queues the time nothing runs.   usage: trace_gaps.py <dir or kernel_trace.csv> [skip_first_n_kernels]
-> how much of a step is launch gap (what a hipGraph of the forward could remove) against kernel time."""
import csv
import glob
import os
import sys


def main():
    p = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if os.path.isdir(p):
        p = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [r for r in csv.DictReader(open(p)) if "asep::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[skip:]
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
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs, rs[1:])]
        gaps = [g for g in gaps if g >= 0]
        if not gaps:
            continue
        gs = sorted(gaps)
        small = [g for g in gaps if g < 20000]
        print(f"  queue {q}: {len(rs)} kernels, gaps median {gs[len(gs) // 2] / 1e3:.2f} us, mean of gaps < 20 us {sum(small) / max(len(small), 1) / 1e3:.2f} us "
              f"({len(small)} of {len(gaps)}), sum of those {sum(small) / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
