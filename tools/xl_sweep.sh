# tuning: the optimizer forms behind the persistent backward, with per-kernel durations (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for cfg in "F0 SWAT_FUSE_ADAMW=0" "PIPE1 SWAT_ADAMW_PIPE=1" "PIPE2 SWAT_ADAMW_PIPE=2" "PIPE1b SWAT_ADAMW_PIPE=1" "PIPE2b SWAT_ADAMW_PIPE=2"; do
  set -- $cfg
  rm -rf $O/sw_$1
  unset SWAT_ADAMW_PIPE SWAT_ADAMW_BLOCKS
  export SWAT_FUSE_ADAMW=1
  tag=$1; shift
  for kv in "$@"; do export $kv; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/sw_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $O/sw_$tag.log 2>&1
  echo "== $cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/sw_$tag.log)"
  (cd $GRAFT_REPO_ROOT && python3 tools/stats_top.py gpurun_out/sw_$tag 25 40 | grep -E "gemm_xl|adamw")
  rm -rf $O/sw_$tag
done
