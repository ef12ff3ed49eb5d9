#!/usr/bin/env python3
"""Busy / idle summary of a rocprofv3 kernel trace (all queues merged): wall span of the last `frac` of the trace, time with at least one
kernel running, the idle gaps by size, and the kernels by total time.  usage: tools/trace_gaps.py <kernel_trace.csv> [frac=0.5]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
t0, end = int(rows[0]["Start_Timestamp"]), int(rows[0]["Start_Timestamp"])
busy, gaps, by = 0, [], collections.Counter()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end:
        gaps.append((s - end, r["Kernel_Name"].split("(")[0][-60:]))
        busy += e - s
    else:
        busy += max(0, e - end)
    end = max(end, e)
    by[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("scldm::", "").replace("void ", "")[:60]] += e - s
span = end - t0
print(f"span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), {len(rows)} kernels, {len(gaps)} idle gaps = {sum(g for g, _ in gaps) / 1e6:.2f} ms")
for lo, hi in ((0, 10e3), (10e3, 50e3), (50e3, 200e3), (200e3, 1e12)):
    sel = [g for g, _ in gaps if lo <= g < hi]
    print(f"  gaps {lo / 1e3:.0f}-{hi / 1e3:.0f} us: {len(sel)} = {sum(sel) / 1e6:.3f} ms")
big = collections.Counter()
for g, name in gaps:
    if g >= 50e3:
        big[name] += g
for name, g in big.most_common(6):
    print(f"  long gaps before {name}: {g / 1e6:.3f} ms")
for name, t in by.most_common(10):
    print(f"  {t / 1e6:8.2f} ms  {name}")
