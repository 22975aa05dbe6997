cd /tmp; export TMPDIR=/tmp
for d in 0 1 2 4 8 3 7 15; do
  export GNSSCORR_ACQ_DBG=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dbg_$d -o acq -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-shared --steps 2 --acq-reps 4 > $GRAFT_REPO_ROOT/gpurun_out/dbg_$d.log 2>&1
  echo "dbg=$d $(grep 'acq_rows3_kernel<true' $GRAFT_REPO_ROOT/gpurun_out/dbg_$d/acq_kernel_stats.csv | awk -F, '{print $4}')"
done
