# tuning: the optimizer forms behind the persistent backward, with per-kernel durations (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for cfg in "G1 SWAT_PIPE_GROUP=1" "G2 SWAT_PIPE_GROUP=2" "G3 SWAT_PIPE_GROUP=3" "G4 SWAT_PIPE_GROUP=4" "G6 SWAT_PIPE_GROUP=6" "G1b SWAT_PIPE_GROUP=1"; do
  set -- $cfg
  rm -rf $O/sw_$1
  unset SWAT_BLOCK_ADAMW SWAT_XL_STAGGER SWAT_PIPE_GROUP
  export SWAT_FUSE_ADAMW=1
  tag=$1; shift
  for kv in "$@"; do export $kv; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/sw_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $O/sw_$tag.log 2>&1
  echo "== $cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/sw_$tag.log)"
  (cd $GRAFT_REPO_ROOT && python3 tools/stats_top.py gpurun_out/sw_$tag 25 40 | grep -E "gemm_xl|adamw")
  rm -rf $O/sw_$tag
done
