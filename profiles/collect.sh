#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers.  Run on the GPU box from the repo root:
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh'
# Outputs land in gpurun_out/prof_*/; profiles/summarise.py turns them into the files committed under profiles/.
# Counter passes are separate runs with --kernel-trace only (no --stats / sys-trace), one counter per pass.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel durations of the whole default bench (tracking + extras + acquisition)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_full" -o full -- python3 "$ROOT/bench.py" --no-cpu --steps 10 > "$OUT/prof_full.log" 2>&1
# 2. tracking kernel alone (the roofline kernel): durations
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_trk" -o trk -- python3 "$ROOT/bench.py" --no-cpu --no-acq --no-shared --steps 20 > "$OUT/prof_trk.log" 2>&1
# 3./4. HBM traffic of the tracking kernel: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_fetch" -o fetch -- python3 "$ROOT/bench.py" --no-cpu --no-acq --no-shared --steps 5 > "$OUT/prof_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/prof_write" -o write -- python3 "$ROOT/bench.py" --no-cpu --no-acq --no-shared --steps 5 > "$OUT/prof_write.log" 2>&1
ls "$OUT"/prof_full "$OUT"/prof_trk "$OUT"/prof_fetch "$OUT"/prof_write
