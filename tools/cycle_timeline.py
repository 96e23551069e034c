#!/usr/bin/env python3
"""From a `rocprofv3 --kernel-trace --output-format csv` trace of tools/cycle_trace.py: the launches of ONE application of the
cycle in order (kernel, grid, duration, gap to the previous launch) -- the last complete run of precond_apply (time_kernel 1).
  python tools/cycle_timeline.py <kernel_trace.csv> [launches per cycle to show]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:48] for r in rows]
# one cycle = from one gather_kernel (entry permutation) to the next
starts = [i for i, n in enumerate(names) if n.startswith("gather_kernel")]
# take a cycle in the middle of the time_kernel(1) loop: consecutive gather starts with identical spacing
best = None
for a, b in zip(starts, starts[1:]):
    if best is None or (b - a) == best[1] - best[0]:
        best = (a, b)
cands = [(a, b) for a, b in zip(starts, starts[1:]) if b - a > 40]
a, b = cands[len(cands) // 2]
t_prev = None
tot = 0
for i in range(a, b):
    r = rows[i]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if t_prev is None else st - t_prev
    t_prev = en
    tot += en - st
    print(f"{i - a:4d} {names[i]:50s} grid={r.get('Grid_Size_X', r.get('Grid_Size', '')):>9s} dur={(en - st) / 1e3:7.1f}us gap={gap / 1e3:6.1f}us")
print("launches", b - a, "kernel time", tot / 1e3, "us, wall", (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3, "us")
