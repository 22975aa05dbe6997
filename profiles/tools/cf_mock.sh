cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GNSSCORR_LIB=$R/gnss-sdr-1_amd/libgnsscorr_exp.so
for d in 0 16 32 48; do
  export GNSSCORR_ACQ_DBG=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_mock_$d -o acq -- python3 $R/profiles/tools/acq_time.py --warm 5 --reps 8 --label mock$d > $R/gpurun_out/r4_mock_$d.log 2>&1
  echo "dbg=$d"; grep -E "rows3_kernel<true|cols_kernel<25, true|mock" $R/gpurun_out/r4_mock_$d/acq_kernel_stats.csv | awk -F'",' '{print $1}' | cut -c1-60 | paste -d' ' - <(grep -E "rows3_kernel<true|cols_kernel<25, true|mock" $R/gpurun_out/r4_mock_$d/acq_kernel_stats.csv | awk -F, '{print $(NF-6), $(NF-4)}')
  grep steady_ms $R/gpurun_out/r4_mock_$d.log | cut -c1-200
done
