#!/bin/bash
# PMC passes over tools/gemm_one.py (run on the GPU box from the repo root): usage  tools/pmc_gemm.sh <tag> <shapes> <cfgs> [exp]
# Three SQ passes (<= 8 counters each) + FETCH_SIZE + WRITE_SIZE; per-kernel means land in gpurun_out/pmc_<tag>.json
ROOT=$GRAFT_REPO_ROOT; TAG=$1; SH=$2; CF=$3; EXP=${4:-0}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1)); rm -rf $ROOT/gpurun_out/pmc_${TAG}_$i
  rocprofv3 --pmc $P --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_$i -- python3 $ROOT/tools/gemm_one.py $SH $CF 6 $EXP > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob("$ROOT/gpurun_out/pmc_${TAG}_*"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void swat::", "")
                res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}
json.dump(out, open("$ROOT/gpurun_out/pmc_${TAG}.json", "w"), indent=1)
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()): print(f"   {c:32s} {v:16.1f}")
PY
