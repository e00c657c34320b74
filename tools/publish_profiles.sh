#!/bin/bash
# Copies what tools/collect_profiles.sh <tag> left under gpurun_out/ into profiles/ under the names profiles/README.md lists.
#   tools/publish_profiles.sh r04        (run here, after the gpurun call has merged gpurun_out/)
TAG=${1:-r04}
cd "$(dirname "$0")/.."
O=gpurun_out
P=profiles
ks() { ls $O/$1/*/*kernel_stats.csv 2>/dev/null | head -1; }
cp "$(ks ${TAG}_prof_stats)" $P/${TAG}_bench_n1_kernel_stats.csv
cp "$(ks ${TAG}_prof_stats0)" $P/${TAG}_bench_n1_overlap0_kernel_stats.csv
tail -1 $O/${TAG}_prof_stats.json > $P/${TAG}_bench_n1_under_rocprof.json
tail -1 $O/${TAG}_prof_stats0.json > $P/${TAG}_bench_n1_overlap0_under_rocprof.json
cp $O/${TAG}_gemm_pmc.json $O/${TAG}_retrieval_pmc.json $P/
cp "$(ks ${TAG}_ret_stats)" $P/${TAG}_retrieval_kernel_stats.csv
cp $O/${TAG}_ret_stats.json $P/${TAG}_retrieval_under_rocprof.json
cp $O/${TAG}_attn_pmc.txt $P/ 2>/dev/null
cp "$(ks ${TAG}_extract_stats)" $P/${TAG}_extract_kernel_stats.csv
cp $O/${TAG}_extract.txt $P/
cp "$(ks ${TAG}_vitl14_stats)" $P/${TAG}_vitl14_overlap0_kernel_stats.csv
cp $O/${TAG}_vitl14.json $P/${TAG}_vitl14_overlap0_under_rocprof.json
cp $O/${TAG}_step_timeline.txt $P/
[ -f $O/${TAG}_bench_full.json ] && grep '^{' $O/${TAG}_bench_full.json | tail -1 > $P/${TAG}_bench_n1_full.json
[ -f $O/${TAG}_mega_bwd_trace.txt ] && cp $O/${TAG}_mega_bwd_trace.txt $P/
[ -f $O/${TAG}_batch_sweep.txt ] && cp $O/${TAG}_batch_sweep.txt $P/
ls -la $P | grep ${TAG}_
