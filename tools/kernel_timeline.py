#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 kernel trace: per kernel start / duration / idle gap since the previous kernel ended
(all queues merged), and the totals.  usage: tools/kernel_timeline.py <kernel_trace.csv> [step index from the end, default 3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the first kernel after the optimizer's launches (torch's multi_tensor_apply group or scldm's adamw_kernel)
opt = lambda r: "multi_tensor_apply" in r["Kernel_Name"] or "adamw_kernel" in r["Kernel_Name"] or "adamw_table_kernel" in r["Kernel_Name"]
starts = [i for i, r in enumerate(rows) if opt(r) and (i + 1 < len(rows) and not opt(rows[i + 1]))]
a, b = starts[-k - 1] + 1, starts[-k] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
end_prev = t0
busy = gap = 0
print(f"{'start us':>9} {'dur us':>8} {'gap us':>7} {'queue':>6}  kernel")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = max(0, s - end_prev)
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("scldm::", "").replace("void ", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {g / 1e3:7.1f} {r.get('Queue_Id', '?'):>6}  {name}")
    gap += g
    end_prev = max(end_prev, e)
print(f"step wall {(end_prev - t0) / 1e3:.1f} us, idle gaps {gap / 1e3:.1f} us, kernels {len(step)}")
