"""Dev tool: one steady-state step of a rocprofv3 kernel trace as a timeline (offset, duration, queue, kernel), plus the
idle gaps of the union over the steps of the timed loop.  python3 tools/step_timeline.py <trace dir> [print_step=1]"""
import csv, glob, sys, collections
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"].split("(")[0].replace("void swat::", "").replace("swat::", "")[:44])
        for r in csv.DictReader(open(path))]
rows.sort()
megas = [r[0] for r in rows if "vit_fwd_mega" in r[3]]
# the timed loop: the longest run of forward launches spaced like back-to-back steps
runs, cur = [], [megas[0]]
for a, b in zip(megas, megas[1:]):
    if b - a < 4.2e6: cur.append(b)
    else: runs.append(cur); cur = [b]
runs.append(cur)
run = max(runs, key=len)
print(f"{len(run)} back-to-back steps, {((run[-1] - run[0]) / (len(run) - 1)) / 1e6:.3f} ms apart")
lo, hi = run[len(run) // 2], run[len(run) // 2 + 1]
step = [r for r in rows if lo <= r[0] < hi]
if len(sys.argv) > 2 and sys.argv[2] == "1":
    for s, e, q, n in step:
        print(f"{(s - lo) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {n}")
# union gaps over the middle half of the run
a, b = run[len(run) // 4], run[3 * len(run) // 4]
win = [r for r in rows if a <= r[0] < b]
nsteps = 3 * len(run) // 4 - len(run) // 4
cur_e, idle, gaps = win[0][1], 0, collections.Counter()
last = win[0]
for r in win[1:]:
    if r[0] > cur_e:
        idle += r[0] - cur_e
        gaps[(last[3], r[3])] += r[0] - cur_e
    if r[1] > cur_e:
        cur_e, last = r[1], r
print(f"per step: {len(win) / nsteps:.0f} launches, kernel sum {sum(e - s for s, e, _, _ in win) / nsteps / 1e6:.3f} ms, idle (no kernel on any queue) {idle / nsteps / 1e6:.3f} ms")
for k, v in gaps.most_common(14):
    print(f"  {v / nsteps / 1e3:7.1f} us/step  {k[0]} -> {k[1]}")
byq = collections.Counter()
for s, e, q, n in win: byq[q] += e - s
print("busy per queue (ms/step):", {q: round(v / nsteps / 1e6, 3) for q, v in byq.items()})
