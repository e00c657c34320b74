#!/bin/bash
# Dev tool: shader clock during the dense candidate scan = GRBM_GUI_ACTIVE / (End - Start) per dispatch, at 200 queries (13 tiles of
# MFMA work per row fragment) and at 16 (one tile): does the scan slow down with the matrix work because the clock does?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/dense_clock
rm -rf $O; mkdir -p $O
NQ=200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/q200 -- python3 $R/tools/perf_dense.py > /dev/null 2>&1
NQ=16 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/q16 -- python3 $R/tools/perf_dense.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("q200", "q16"):
    rows = collections.defaultdict(dict)
    for f in glob.glob("$O/" + p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "shadow2" not in r["Kernel_Name"]: continue
            e = rows[r["Dispatch_Id"]]
            e[r["Counter_Name"]] = float(r["Counter_Value"]); e["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); e["k"] = r["Kernel_Name"].split("(")[0][-45:]
    v = list(rows.values())[2:]
    if not v: print(p, "no dispatches"); continue
    n = len(v)
    ns = sum(e["ns"] for e in v) / n; gui = sum(e["GRBM_GUI_ACTIVE"] for e in v) / n; mf = sum(e["SQ_VALU_MFMA_BUSY_CYCLES"] for e in v) / n
    print(f"{p} {v[0]['k']}: {n} launches, {ns / 1e3:.1f} us, GRBM_GUI_ACTIVE {gui:.0f} -> {gui / ns:.3f} GHz; MFMA busy {mf / 1024:.0f} cycles per SIMD = {mf / 1024 / gui:.3f} of the launch")
PY
