#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers.  Run on the GPU box from the repo root:
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh'
# Outputs land in gpurun_out/prof_*/; profiles/summarise.py turns them into the files committed under profiles/.
# Counter passes are separate runs with --kernel-trace only (no --stats / sys-trace), one counter group per pass.
# PASSES (default: all) selects a subset, e.g. PASSES="acq" bash profiles/collect.sh
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
PASSES="${PASSES:-line full trk trksq acq sq}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
has() { case " $PASSES " in *" $1 "*) return 0;; *) return 1;; esac; }
if has line; then
# 0. the bench line exactly as the driver asks for it (no profiler attached)
python3 "$ROOT/bench.py" --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
fi
if has full; then
# 1. per-kernel durations of the whole default bench (tracking + extras + acquisition)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_full" -o full -- python3 "$ROOT/bench.py" --no-cpu --steps 10 > "$OUT/prof_full.log" 2>&1
fi
if has trk; then
# 2. tracking kernel alone (the roofline kernel): durations
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_trk" -o trk -- python3 "$ROOT/bench.py" --no-cpu --no-acq --no-shared --steps 20 --warmup 5 > "$OUT/prof_trk.log" 2>&1
# 3./4. HBM traffic of the tracking kernel: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_fetch" -o fetch -- python3 "$ROOT/bench.py" --no-cpu --no-acq --no-shared --preroll-ms 0 --segments 0 --steps 5 > "$OUT/prof_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/prof_write" -o write -- python3 "$ROOT/bench.py" --no-cpu --no-acq --no-shared --preroll-ms 0 --segments 0 --steps 5 > "$OUT/prof_write.log" 2>&1
fi
if has trksq; then
# 4b. where the instructions of the tracking kernels go (every mode of the bench line): SQ counters per kernel name
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$OUT/prof_trk_sq" -o sq -- python3 "$ROOT/bench.py" --no-cpu --no-acq --preroll-ms 0 --segments 0 --steps 3 > "$OUT/prof_trk_sq.log" 2>&1
fi
if has acq; then
# 5. acquisition (cfg4: 32 PRNs x 41 bins x 2 dwells): per-kernel durations, then FETCH_SIZE / WRITE_SIZE of every acq_* kernel.
#    bench.py runs 1 warm-up + --acq-reps timed searches; summarise.py divides by that count (the 32 set_local_code transforms
#    of the set-up use the same kernels on one cell each: < 1 % of a search's traffic)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_acq" -o acq -- python3 "$ROOT/bench.py" --no-cpu --no-shared --steps 3 --preroll-ms 0 --segments 0 --acq-warmup 1 --acq-reps 9 > "$OUT/prof_acq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_acq_fetch" -o fetch -- python3 "$ROOT/bench.py" --no-cpu --no-shared --steps 3 --preroll-ms 0 --segments 0 --acq-warmup 1 --acq-reps 4 > "$OUT/prof_acq_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/prof_acq_write" -o write -- python3 "$ROOT/bench.py" --no-cpu --no-shared --steps 3 --preroll-ms 0 --segments 0 --acq-warmup 1 --acq-reps 4 > "$OUT/prof_acq_write.log" 2>&1
fi
if has sq; then
# 6. where the instructions of the acquisition kernels go: SQ counters (8 slots per pass)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/prof_acq_sq1" -o sq1 -- python3 "$ROOT/bench.py" --no-cpu --no-shared --steps 3 --preroll-ms 0 --segments 0 --acq-warmup 1 --acq-reps 4 > "$OUT/prof_acq_sq1.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$OUT/prof_acq_sq2" -o sq2 -- python3 "$ROOT/bench.py" --no-cpu --no-shared --steps 3 --preroll-ms 0 --segments 0 --acq-warmup 1 --acq-reps 4 > "$OUT/prof_acq_sq2.log" 2>&1
fi
ls "$OUT"/prof_*/ | head -80
