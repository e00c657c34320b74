#!/bin/bash
# Dev tool: instruction-cache counters of the dense candidate scan (is the unrolled candidate epilogue thrashing the 64 KB I-cache?)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/dense_icache
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_BUSY_CYCLES --output-format csv -d $O/p -- python3 $R/tools/ab_dense_hits.py > $O/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("$O/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "shadow2" not in r["Kernel_Name"]: continue
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
print(len(ids), "scan launches")
# ab_dense_hits: 1 warm call, then 5 settings x 8 reps x 5 launches
body = ids[1:]
for i, name in enumerate(("real", "+0.01", "+0.02", "+0.04", "none")):
    grp = body[i * 40:(i + 1) * 40]
    if not grp: break
    avg = {k: sum(rows[d].get(k, 0) for d in grp) / len(grp) for k in rows[grp[0]]}
    print(name, "  ".join(f"{k}={v:.4g}" for k, v in sorted(avg.items())))
PY
tail -5 $O/log.txt
