set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --no-cpu --no-shared --steps 3 --preroll-ms 0 --segments 0 --acq-warmup 1 --acq-reps 4"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_stats -o s -- $B > $R/gpurun_out/pf_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pf_sq1 -o sq1 -- $B > $R/gpurun_out/pf_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pf_sq2 -o sq2 -- $B > $R/gpurun_out/pf_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pf_fetch -o f -- $B > $R/gpurun_out/pf_fetch.log 2>&1
