"""timeline summary of a rocprofv3 --kernel-trace/--memory-copy-trace directory: the LAST host call's kernels and copies in
start order with gaps.  Usage: python tools/summarize_trace.py <dir>"""
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
# the last call = everything after the last gap > 300 us ... walk back from the end
end = len(rows)
i = end - 1
while i > 0 and rows[i][0] - max(r[1] for r in rows[max(0, i - 40):i]) < 300_000:
    i -= 1
t0 = rows[i][0]
last_end = t0
for s, e, n in rows[i:]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {(s - last_end) / 1e3:7.1f}  {n}")
    last_end = max(last_end, e)
print("span", (last_end - t0) / 1e3, "us")
