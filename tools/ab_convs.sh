#!/bin/bash
# the window's convs in one operand format under several A/B libraries (tools/build_ab.sh), interleaved: $1 = dtype, rest = library names ("-" = product)
dt=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset HVR_BENCH_LIB; else export HVR_BENCH_LIB=abtest/libhvr_$lib.so; fi
    echo "== round $round lib $lib"
    python tools/conv_hint_sweep.py --dtype $dt --hints 11 2>&1 | grep -v amdgpu | awk '{printf "%s %s %s | ", $1, $2, $(NF)} END {print ""}'
  done
done
