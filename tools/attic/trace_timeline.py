#!/usr/bin/env python3
"""Average timeline of a repeated kernel sequence from a rocprofv3 --kernel-trace csv: per position in the period, the kernel's
name, its average duration and the average idle gap in front of it.   usage: trace_timeline.py <kernel_trace.csv> [skip_iters]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
# period = distance between occurrences of the LAST kernel's name pattern: find the smallest p with names[-p:] == names[-2p:-p]
period = None
for p in range(1, len(names) // 3):
    if names[-p:] == names[-2 * p:-p] == names[-3 * p:-2 * p]:
        period = p
        break
if period is None:
    sys.exit("no repeating kernel sequence found")
iters = len(names) // period - skip
tail = rows[-iters * period:]
tot = 0.0
print(f"period {period} kernels, {iters} iterations averaged")
for k in range(period):
    dur = gap = 0.0
    for i in range(iters):
        r = tail[i * period + k]
        dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if i * period + k > 0:
            prev = tail[i * period + k - 1]
            gap += int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])
    dur /= iters * 1e3
    gap /= iters * 1e3
    tot += dur + gap
    print(f"{k:3d} gap {gap:7.1f} us  run {dur:7.1f} us  {tail[k]['Kernel_Name'][:110]}")
print(f"sum of runs + gaps per iteration: {tot:.1f} us")
