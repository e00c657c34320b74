#!/bin/bash
# usage: tools/pmc_gemm.sh <layout> <M> <N> <K> <cfg>   (on the GPU box, from the repo root)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$1; M=$2; N=$3; K=$4; C=$5
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_a -- python3 tools/one_gemm.py $L $M $N $K $C 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_b -- python3 tools/one_gemm.py $L $M $N $K $C 0 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_a", "pmc_b"):
    f = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]) / 5
    print({k: round(v) for k, v in agg.items()})
f = glob.glob("gpurun_out/pmc_a/*/*kernel_trace.csv")[0]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
print("kernel ns", d)
PY
