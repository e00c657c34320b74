#!/bin/bash
# Dev tool: SQ counters of the dense top-k's candidate scan (sim_candidates_shadow2_kernel) over tools/perf_dense.py, two passes:
# how busy the matrix pipe and the LDS are while the scan streams the shard (round 5: the scan is co-limited by both at 200 queries).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/dense_pmc
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY \
    --output-format csv -d $O/p1 -- python3 $R/tools/perf_dense.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_SALU \
    --output-format csv -d $O/p2 -- python3 $R/tools/perf_dense.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/" + p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60] + " g" + r["Grid_Size"]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(acc.items()):
        if "shadow2" not in k: continue
        print(k, "launches", len(next(iter(c.values()))))
        print("   " + "  ".join(f"{n}={sum(v) / len(v):.4g}" for n, v in sorted(c.items())))
        if "SQ_BUSY_CYCLES" in c:
            cyc = sum(c["SQ_BUSY_CYCLES"]) / len(c["SQ_BUSY_CYCLES"]) / 32.0
            print(f"   kernel cycles {cyc:.0f}; mfma busy {sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / len(c['SQ_VALU_MFMA_BUSY_CYCLES']) / (1024.0 * cyc):.3f} of the SIMDs' cycles; "
                  f"LDS instructions active {sum(c['SQ_ACTIVE_INST_LDS']) / len(c['SQ_ACTIVE_INST_LDS']) / sum(c['SQ_WAVE_CYCLES']) * len(c['SQ_WAVE_CYCLES']):.3f} and waiting {sum(c['SQ_WAIT_INST_LDS']) / sum(c['SQ_WAVE_CYCLES']):.3f} of wave cycles; wait any {sum(c['SQ_WAIT_ANY']) / sum(c['SQ_WAVE_CYCLES']):.3f}")
PY
