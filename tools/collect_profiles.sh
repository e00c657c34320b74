#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline object cites (run on the GPU box from the repo root):
#   tools/collect_profiles.sh <tag>      e.g. r02   -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
#   <tag>_prof_stats / _prof_stats0 : --kernel-trace --stats of the N=1 bench command (default / --overlap 0)
#   <tag>_pmc_sq                    : SQ counters per GEMM instantiation (MFMA busy, waits, LDS conflicts)
#   <tag>_pmc_fetch, _pmc_write     : FETCH_SIZE / WRITE_SIZE in separate passes (MI355X_MICROARCH.md §HBM)
# Counter passes never combine --pmc with a trace domain; the program sits directly behind `--`.
TAG=${1:-r02}
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-secondary"
O=$ROOT/gpurun_out
rm -rf $O/${TAG}_prof_stats $O/${TAG}_prof_stats0 $O/${TAG}_pmc_sq $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_stats -- $BENCH --steps 20 --warmup 5 > $O/${TAG}_prof_stats.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_stats0 -- $BENCH --steps 20 --warmup 5 --overlap 0 > $O/${TAG}_prof_stats0.json 2> /dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
    --output-format csv -d $O/${TAG}_pmc_sq -- $BENCH --steps 4 --warmup 2 --overlap 0 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- $BENCH --steps 4 --warmup 2 --overlap 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- $BENCH --steps 4 --warmup 2 --overlap 0 > /dev/null 2>&1
python3 - <<PY
import csv, glob, json, collections
O, TAG = "$O", "$TAG"
def load(d):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{TAG}_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void swat::", "")
                res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return res
sq, fe, wr = load("pmc_sq"), load("pmc_fetch"), load("pmc_write")
out = {"source": "rocprofv3 --pmc (three separate passes: 8 SQ counters; FETCH_SIZE; WRITE_SIZE) over `bench.py --steps 4 --warmup 2 --overlap 0 "
                 "--no-cpu-baseline --no-secondary` (kernels run alone, one stream); means per launch of every GEMM instantiation",
       "reading": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 (the counter sums "
                  "the 32 shader engines); wait fractions are of SQ_WAVE_CYCLES (quad-cycles); FETCH_SIZE is doubled (gfx950 tallies 128-B "
                  "requests at 64 B, MI355X_MICROARCH.md HBM section), WRITE_SIZE as is; KB -> bytes",
       "kernels": {}}
tot_f = tot_w = n_f = n_w = 0
for k in sorted(sq):
    c = {n: sum(v) / len(v) for n, v in sq[k].items()}
    e = {"launches_seen": len(next(iter(sq[k].values()))), "counters": c}
    cyc = c.get("SQ_BUSY_CYCLES", 0) / 32.0
    if cyc:
        e["kernel_cycles"] = cyc
        e["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * cyc)
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
            e[n.lower() + "_frac_of_wave_cycles"] = c.get(n, 0) / wc
    if k in fe:
        v = fe[k]["FETCH_SIZE"]; e["hbm_fetch_bytes"] = 2 * 1024 * sum(v) / len(v); tot_f += 2 * 1024 * sum(v); n_f += len(v)
    if k in wr:
        v = wr[k]["WRITE_SIZE"]; e["hbm_write_bytes"] = 1024 * sum(v) / len(v); tot_w += 1024 * sum(v); n_w += len(v)
    out["kernels"][k] = e
if n_f and n_w:
    out["hbm_bytes_per_gemm_launch"] = tot_f / n_f + tot_w / n_w
    out["gemm_launches"] = n_f
json.dump(out, open(f"{O}/{TAG}_gemm_pmc.json", "w"), indent=1)
for k, e in out["kernels"].items():
    print(f"{k:52s} mfma busy {e.get('mfma_busy_frac', 0):.3f}  wait_any {e.get('sq_wait_any_frac_of_wave_cycles', 0):.2f}  "
          f"lds conflicts {e['counters'].get('SQ_LDS_BANK_CONFLICT', 0):.0f}  fetch {e.get('hbm_fetch_bytes', 0) / 1e6:.1f} MB  write {e.get('hbm_write_bytes', 0) / 1e6:.1f} MB")
print("hbm bytes per GEMM launch:", out.get("hbm_bytes_per_gemm_launch"))
PY
tail -1 $O/${TAG}_prof_stats.json | cut -c1-300
cd $ROOT && python3 tools/stats_top.py gpurun_out/${TAG}_prof_stats0 25 22
