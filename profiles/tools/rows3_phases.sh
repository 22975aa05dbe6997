# Phase-elimination timing of acq_rows3_kernel<true>: build the experiments library with the switches compiled in
#   make -C gnss-sdr-1_amd/csrc exp EXTRA=-DACQ_ROWS3_DBG=1
# let it travel (drop its line from .gpurunignore for the call) and run this on the GPU box (gpurun -- 'bash profiles/tools/rows3_phases.sh').
# Results are wrong by construction with any switch set; only the times are read: whole search (acq_time.py) and, under
# rocprofv3, the row kernel's own duration.  bits: 1 = no global input loads, 2 = no global stores, 4 = no twiddle loads, 8 = no butterflies
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GNSSCORR_LIB=$R/gnss-sdr-1_amd/libgnsscorr_exp.so
for d in 0 1 2 4 8 3 7 15; do
  export GNSSCORR_ACQ_DBG=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_dbg_$d -o acq -- python3 $R/profiles/tools/acq_time.py --warm 5 --reps 8 --label dbg$d > $R/gpurun_out/r4_dbg_$d.log 2>&1
  echo "dbg=$d rows3<true> avg ns: $(grep 'acq_rows3_kernel<true' $R/gpurun_out/r4_dbg_$d/acq_kernel_stats.csv | awk -F, '{print $4}')  $(grep steady_ms $R/gpurun_out/r4_dbg_$d.log | head -1 | cut -c1-160)"
done
