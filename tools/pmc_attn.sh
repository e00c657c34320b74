#!/bin/bash
# Dev tool: SQ counters of the tiled attention kernels (tools/perf_attn.py), two passes, averaged per kernel.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/attn_pmc
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY \
    --output-format csv -d $O/p1 -- python3 $R/tools/perf_attn.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAVES SQ_INSTS_SALU \
    --output-format csv -d $O/p2 -- python3 $R/tools/perf_attn.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/" + p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60] + " g" + r["Grid_Size"]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(acc.items()):
        if "attn" not in k: continue
        print(k)
        print("   " + "  ".join(f"{n}={sum(v) / len(v):.3g}" for n, v in sorted(c.items())))
PY
