#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$1; M=$2; N=$3; K=$4; C=$5
rm -rf gpurun_out/pmc_c
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc_c -- python3 tools/one_gemm.py $L $M $N $K $C 0 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_c/*/*counter_collection.csv")[0]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "gemm" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]) / 5
print("cfg", $C, {k: round(v) for k, v in agg.items()}, "hit rate %.3f" % (agg["TCC_HIT_sum"] / max(1, agg["TCC_HIT_sum"] + agg["TCC_MISS_sum"])), "EA read MB %.1f" % (agg["TCC_EA0_RDREQ_sum"] * 64 / 1e6))
PY
