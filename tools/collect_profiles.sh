#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline object cites (run on the GPU box from the repo root):
#   gpurun_out/prof_stats : --kernel-trace --stats of the N=1 bench command
#   gpurun_out/pmc_fetch, pmc_write : FETCH_SIZE / WRITE_SIZE in separate passes (MI355X_MICROARCH.md §HBM)
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
rm -rf $ROOT/gpurun_out/prof_stats $ROOT/gpurun_out/pmc_fetch $ROOT/gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_stats -- $CMD > $ROOT/gpurun_out/prof_stats.json 2> /dev/null
# same command with the backward's lanes serialised (--overlap 0): per-kernel durations of kernels running alone, the
# figure bench.py's roofline object reports (its instrumented replay also runs the lanes serialised)
rm -rf $ROOT/gpurun_out/prof_stats0
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_stats0 -- $CMD --overlap 0 > $ROOT/gpurun_out/prof_stats0.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_fetch -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_write -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 - <<PY
import csv, glob, json
out = {}
for name, d in (("fetch", "pmc_fetch"), ("write", "pmc_write")):
    f = glob.glob("$ROOT/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if "gemm_" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
    out[name] = (tot, n)
(fs, n1), (wsz, n2) = out["fetch"], out["write"]
res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 6 --warmup 2 (8 + 6 replayed steps), all gemm_* kernels",
       "gemm_launches": n1, "fetch_size_kb_sum": fs, "write_size_kb_sum": wsz,
       "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section); WRITE_SIZE as is",
       "hbm_bytes_per_gemm_launch": (2 * fs * 1024 / n1 + wsz * 1024 / n2)}
json.dump(res, open("$ROOT/gpurun_out/gemm_hbm_traffic.json", "w"), indent=1)
print(res)
PY
tail -1 $ROOT/gpurun_out/prof_stats.json | cut -c1-300
ls $ROOT/gpurun_out/prof_stats/*/ | head
