#!/bin/bash
# Builds gnss-sdr-1_amd/libgnsscorr_<name>.so = the product library with acq_kernels.hip (and optionally other sources) recompiled with extra
# flags, for A/B timing through $GNSSCORR_LIB (profiles/tools/acq_time.py).  Usage: build_variant.sh <name> "<flags>" [source.hip ...]
set -e
NAME=$1; FLAGS=$2; shift 2
SRCS=${@:-acq_kernels.hip}
cd "$(dirname "$0")/../../gnss-sdr-1_amd/csrc"
mkdir -p var_obj
OBJS=""
for f in gc_context gc_stream gc_tracking trk_kernels trk_closed_loop gc_acquisition acq_kernels gc_codes; do
  o=$f.o
  for s in $SRCS; do
    if [ "$s" = "$f.hip" ]; then
      o=var_obj/${f}_$NAME.o
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wall -Wno-unused-function $FLAGS -c $s -o $o
    fi
  done
  OBJS="$OBJS $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgnsscorr_$NAME.so $OBJS -ldl
echo built ../libgnsscorr_$NAME.so
