#!/bin/bash
# Builds libhvr_hip.so (gfx950 only) in-tree next to the sources.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libhvr_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
mkdir -p build
pids=()
for f in gemm misc roi_align nms stem capi; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ gemm_params.h -nt build/$f.o ] || [ ../../include/hvr_hip.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/gemm.o build/misc.o build/roi_align.o build/nms.o build/stem.o build/capi.o -o $OUT
echo "built $(realpath $OUT)"
