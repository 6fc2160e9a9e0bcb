#!/usr/bin/env python3
"""Timeline of the LAST rsx_sc_query host call in a rocprofv3 csv trace (kernel + memory-copy trace): every kernel and copy
with its start relative to the call's first event.  Usage: host_timeline.py <dir with *kernel_trace.csv, *memory_copy_trace.csv>"""
import csv, glob, re, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"([A-Za-z_0-9]+)\(", n)
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + (m.group(1) if m else n)[:34]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "?")[:30]))
ev.sort()
# the last call = events after the last gap of more than 300 us ... walk back from the end
i = len(ev) - 1
while i > 0 and ev[i][0] - max(e[1] for e in ev[max(0, i - 6):i]) < 300_000:
    i -= 1
call = ev[i:]
t0 = call[0][0]
for s, e, n in call:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  {n}")
print(f"span {(max(e for _, e, _ in call) - t0) / 1e3:.1f} us, {len(call)} events")
