"""Summarise a rocprofv3 kernel trace: busy union, idle gaps, per-kernel totals over the last steps (tuning aid)."""
import csv, sys, glob, collections
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
# last 40 % of the trace (steady state)
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * 6 // 10
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = collections.Counter()
gapn = collections.Counter()
prev = rows[0]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps[(prev[2], n)] += s - cur_e
        gapn[(prev[2], n)] += 1
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        prev = (s, e, n)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
print(f"span {span/1e6:.3f} ms  busy-union {busy/1e6:.3f} ms ({100*busy/span:.1f} %)  sum of kernels {tot/1e6:.3f} ms  launches {len(rows)}")
print("largest idle-gap classes (total us, count, before -> after):")
for k, v in gaps.most_common(12):
    print(f"  {v/1e3:9.1f} {gapn[k]:5d}  {k[0]} -> {k[1]}")
agg = collections.Counter(); cnt = collections.Counter()
for s, e, n in rows:
    agg[n] += e - s; cnt[n] += 1
print("kernel totals (us, count, avg us):")
for k, v in agg.most_common(16):
    print(f"  {v/1e3:9.1f} {cnt[k]:5d} {v/1e3/cnt[k]:8.2f}  {k}")
