"""Top rows of a rocprofv3 kernel_stats.csv: python3 tools/stats_top.py <dir> <steps>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("sum of kernels per step (ms)", round(tot / n / 1e6, 3))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print("%-60s %7.1f %9.1f us %7.3f ms" % (r["Name"][:60], int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / n / 1e6))
