# A/B of library variants (profiles/tools/build_variant.sh): per variant the row / column kernels' mean durations under rocprofv3 and the
# search time without the profiler.  Usage on the GPU box: VARIANTS="h1 h0" bash profiles/tools/ab_variants.sh
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in $VARIANTS; do
  export GNSSCORR_LIB=$R/gnss-sdr-1_amd/libgnsscorr_$v.so
  python3 $R/profiles/tools/acq_time.py --label "$v" 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['label'], 'cold %.4f steady %.4f found %s' % (j['cold_ms'], j['steady_ms'], j['found']))"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$v -o acq -- python3 $R/profiles/tools/acq_time.py --warm 5 --reps 10 > /dev/null 2>&1
  grep -E "rows3_kernel<true|cols_kernel<25, true" $R/gpurun_out/ab_$v/acq_kernel_stats.csv | awk -F, '{n=split($0,a,","); print "   ", substr($1,1,40), a[n-4]}'
done
done
