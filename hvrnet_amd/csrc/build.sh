#!/bin/bash
# Builds libhvr_hip.so (gfx950 only) in-tree next to the sources.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libhvr_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
mkdir -p build
pids=()
for f in gemm gemm_f16 expand expand_split conv3x3 pc_gemm bigtile kpar misc roi_align nms stem relation_bt relation_apply_bt targets ingest capi; do
  need_remarks=0
  if { [ $f = gemm ] || [ $f = gemm_f16 ] || [ $f = expand ] || [ $f = nms ] || [ $f = relation_bt ] || [ $f = relation_apply_bt ] || [ $f = pc_gemm ] || [ $f = bigtile ] || [ $f = conv3x3 ] || [ $f = expand_split ] || [ $f = kpar ]; } && [ ! -f build/$f.remarks ]; then need_remarks=1; fi
  if [ $need_remarks = 1 ] || [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ gemm_params.h -nt build/$f.o ] || [ gemm_tile.h -nt build/$f.o ] || [ relation_bt.h -nt build/$f.o ] || [ ../../include/hvr_hip.h -nt build/$f.o ]; then
    if [ $f = gemm ] || [ $f = gemm_f16 ] || [ $f = expand ] || [ $f = nms ] || [ $f = relation_bt ] || [ $f = relation_apply_bt ] || [ $f = pc_gemm ] || [ $f = bigtile ] || [ $f = conv3x3 ] || [ $f = expand_split ] || [ $f = kpar ]; then
      # (-save-temps=obj: the device assembly build/$f-hip-amdgcn-amd-amdhsa-gfx950.s, which check_asm_waits.py scans)
      ( hipcc $FLAGS -save-temps=obj -Rpass-analysis=kernel-resource-usage -c $f.hip -o build/$f.o 2> build/$f.remarks; rm -f build/$f-h*.hipi build/$f-h*.bc build/$f-host-*.s build/$f-hip-*.out* ) &
    elif [ $f = targets ] || [ $f = ingest ]; then  # thresholds / equality tests on IoUs: keep the reference's rounding (no fused multiply-add)
      hipcc $FLAGS -ffp-contract=off -c $f.hip -o build/$f.o &
    else
      hipcc $FLAGS -c $f.hip -o build/$f.o &
    fi
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
for f in gemm gemm_f16 expand nms relation_bt relation_apply_bt pc_gemm bigtile conv3x3 expand_split kpar; do
  if grep -q "error:" build/$f.remarks 2>/dev/null; then grep -A3 "error:" build/$f.remarks; rm -f build/$f.o; exit 1; fi
done
python3 check_asm_waits.py build/gemm-hip-amdgcn-amd-amdhsa-gfx950.s build/gemm_f16-hip-amdgcn-amd-amdhsa-gfx950.s build/expand-hip-amdgcn-amd-amdhsa-gfx950.s build/relation_bt-hip-amdgcn-amd-amdhsa-gfx950.s build/relation_apply_bt-hip-amdgcn-amd-amdhsa-gfx950.s build/pc_gemm-hip-amdgcn-amd-amdhsa-gfx950.s build/bigtile-hip-amdgcn-amd-amdhsa-gfx950.s build/conv3x3-hip-amdgcn-amd-amdhsa-gfx950.s build/expand_split-hip-amdgcn-amd-amdhsa-gfx950.s build/kpar-hip-amdgcn-amd-amdhsa-gfx950.s
python3 check_regs.py build/expand_split.remarks build/gemm.remarks build/gemm_f16.remarks build/expand.remarks build/nms.remarks build/relation_bt.remarks build/relation_apply_bt.remarks build/pc_gemm.remarks build/bigtile.remarks build/kpar.remarks
hipcc --offload-arch=gfx950 -shared -fPIC build/gemm.o build/gemm_f16.o build/expand.o build/expand_split.o build/conv3x3.o build/pc_gemm.o build/bigtile.o build/kpar.o build/misc.o build/roi_align.o build/nms.o build/stem.o build/relation_bt.o build/relation_apply_bt.o build/targets.o build/ingest.o build/capi.o -o $OUT
echo "built $(realpath $OUT)"
