#!/bin/bash
# (GEMM_SAME=1: keep gemm.o; EXPAND_DBG=1: expand.hip recompiled with the defines too)
# tuning builds: tools/build_dbg.sh NAME "-DFOO ..."  ->  dbg/libhvr_NAME.so (gemm.hip recompiled with the extra defines)
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; defs=${2:-}
mkdir -p dbg/$name
B=hvrnet_amd/csrc/build
[ -n "${GEMM_SAME:-}" ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $defs -c hvrnet_amd/csrc/gemm.hip -o dbg/$name/gemm.o
if [ -n "${GEMM_SAME:-}" ]; then cp $B/gemm.o dbg/$name/gemm.o; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $defs -c hvrnet_amd/csrc/relation_bt.hip -o dbg/$name/relation_bt.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $defs -c hvrnet_amd/csrc/relation_apply_bt.hip -o dbg/$name/relation_apply_bt.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $defs -c hvrnet_amd/csrc/pc_gemm.hip -o dbg/$name/pc_gemm.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DHVR_DEBUG_KNOBS $defs -c hvrnet_amd/csrc/capi.hip -o dbg/$name/capi.o
N=$B/nms.o
if [ -n "${NMS_DBG:-}" ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $defs -c hvrnet_amd/csrc/nms.hip -o dbg/$name/nms.o; N=dbg/$name/nms.o; fi
X=$B/expand.o
if [ -n "${EXPAND_DBG:-}" ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $defs -c hvrnet_amd/csrc/expand.hip -o dbg/$name/expand.o; X=dbg/$name/expand.o; fi
# (gemm_f16.o / bigtile.o: the product build's objects -- half-operand tile kernels and the 288 x 256 tiles are not what these builds probe)
hipcc --offload-arch=gfx950 -shared -fPIC dbg/$name/gemm.o $B/gemm_f16.o $B/bigtile.o $B/kpar.o $X $B/conv3x3.o dbg/$name/pc_gemm.o $B/misc.o $B/roi_align.o $N $B/stem.o $B/targets.o $B/ingest.o dbg/$name/relation_apply_bt.o $B/expand_split.o dbg/$name/relation_bt.o dbg/$name/capi.o -o dbg/libhvr_$name.so
rm -rf dbg/$name
echo built dbg/libhvr_$name.so
