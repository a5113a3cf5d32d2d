#!/usr/bin/env python3
"""How much of a rocprofv3 --kernel-trace of the library's kernels ran concurrently: sum of kernel durations vs the union of
their intervals, per kernel name and in all.   python tools/bench_extra/overlap_of_trace.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].lstrip("void ").startswith("k_")]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "")) for r in rows)
tot = sum(e - s for s, e, _, _ in iv)
union, cur_s, cur_e = 0, None, None
for s, e, _, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"kernels {len(iv)}  sum {tot / 1e6:.1f} ms  union {union / 1e6:.1f} ms  span {(iv[-1][1] - iv[0][0]) / 1e6:.1f} ms  queues {sorted(set(q for *_, q in iv))}")
by = defaultdict(lambda: [0, 0])
for s, e, n, _ in iv:
    by[n][0] += 1
    by[n][1] += e - s
for n, (c, t) in sorted(by.items(), key=lambda x: -x[1][1])[:12]:
    print(f"  {n:40s} {c:6d} {t / 1e6:9.2f} ms  avg {t / c / 1e3:9.1f} us")
