#!/bin/bash
# Variant builds for probes: tools/build_var.sh NAME "-DFOO ..." file1 [file2 ...]  ->  abtest/libhvr_NAME.so (abtest/ travels to the GPU box, dbg/ does not)
# The listed sources (names without .hip, e.g. "bigtile expand") are recompiled with the extra defines; every other object is the
# product build's (hvrnet_amd/csrc/build/*.o).  Results of ablation builds (-DHVR_DBG_X_*) are timings only, never valid outputs.
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; defs=$2; shift 2
B=hvrnet_amd/csrc/build
mkdir -p dbg/$name
objs=()
for f in gemm gemm_f16 expand expand_split conv3x3 pc_gemm bigtile kpar misc roi_align nms stem relation_bt relation_apply_bt targets ingest capi; do
  o=$B/$f.o
  for v in "$@"; do
    if [ "$v" = "$f" ]; then
      extra=""
      if [ $f = targets ] || [ $f = ingest ]; then extra="-ffp-contract=off"; fi
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $extra $defs -c hvrnet_amd/csrc/$f.hip -o dbg/$name/$f.o &
      o=dbg/$name/$f.o
    fi
  done
  objs+=($o)
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o abtest/libhvr_$name.so
rm -rf dbg/$name
echo built abtest/libhvr_$name.so
