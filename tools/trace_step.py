#!/usr/bin/env python3
"""Per-step kernel timeline from a rocprofv3 kernel trace: wall span of the last step's backward kernels and overlaps.
usage: trace_step.py <kernel_trace.csv> [name of the step's first kernel]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last occurrence of the sort's first kernel marks the last step
first_kernel = sys.argv[2] if len(sys.argv) > 2 else 'psort_count'
starts = [i for i, r in enumerate(rows) if first_kernel in r['Kernel_Name']]
last = rows[starts[-2]:starts[-1]] if len(starts) > 1 else rows[starts[-1]:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void shacira::', '').replace('shacira::', '')[:44]
    print(f"{name:44s} start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f}")
print("step span us:", (max(int(r['End_Timestamp']) for r in last) - t0) / 1e3)
