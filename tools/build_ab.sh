#!/bin/bash
# A/B build of the tile engine with extra defines:  tools/build_ab.sh NAME "-DFOO=1"  ->  abtest/libhvr_NAME.so
# (gemm.hip and gemm_f16.hip recompiled with the defines, every other object from the product build; abtest/ travels to the GPU box --
# *.so is git-ignored, not gpurun-ignored -- and tools/*.py pick a library with HVR_BENCH_LIB=abtest/libhvr_NAME.so)
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; defs=${2:-}
mkdir -p abtest/$name
B=hvrnet_amd/csrc/build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
hipcc $F $defs -c hvrnet_amd/csrc/gemm.hip -o abtest/$name/gemm.o &
hipcc $F $defs -c hvrnet_amd/csrc/gemm_f16.hip -o abtest/$name/gemm_f16.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC abtest/$name/gemm.o abtest/$name/gemm_f16.o $B/expand.o $B/expand_split.o $B/conv3x3.o $B/pc_gemm.o $B/bigtile.o $B/misc.o $B/roi_align.o $B/nms.o $B/stem.o $B/relation_bt.o $B/targets.o $B/ingest.o $B/capi.o -o abtest/libhvr_$name.so
rm -rf abtest/$name
echo built abtest/libhvr_$name.so
