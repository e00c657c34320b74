"""Dev tool: a time window of a rocprofv3 kernel trace (offset, duration, queue, kernel) — for steps without the persistent forward's
launch as a marker.  python3 tools/trace_window.py <trace dir> <start fraction 0..1> <window us>"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"].split("(")[0].replace("void swat::", "").replace("swat::", "")[:40])
        for r in csv.DictReader(open(path))]
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
lo = rows[int(len(rows) * float(sys.argv[2]))][0]   # (a fraction of the launches, not of the time span: idle stretches hold no rows)
hi = lo + int(float(sys.argv[3]) * 1e3)
for s, e, q, n in rows:
    if lo <= s < hi:
        print(f"{(s - lo) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {n}")
