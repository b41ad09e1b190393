#!/bin/bash
# Every tuning knob of the library in its non-default position, one at a time, through the smoke test (a small window against the CPU
# oracle: classes exact, boxes / scores < 1e-3) and a short graph-replay run (stream + clip graphs): the alternate paths get no other
# regular exercise, and one of them hid a GPU memory fault until round 4 (profiles/r04_graph_memset_fault.txt).
#   gpurun --timeout 1800 -- 'bash tools/knob_sweep.sh > gpurun_out/knob_sweep.txt 2>&1'
cd ${GRAFT_REPO_ROOT:-.}
knobs="HVR_BIGTILE=0 HVR_BIGTILE_MIN=64 HVR_BIGTILE_RES=0 HVR_BIGTILE_RES_SHARED=1 HVR_CONV3=0 HVR_CONV_SPLITK=0 HVR_EXPAND=0 HVR_FRAME_GROUPS=2 HVR_FUSE_NEXT=0 HVR_FUSE_TAIL=0 HVR_GM_APPLY=4 HVR_GM_SCORES=1 HVR_KEY_MERGE=1 HVR_KEY_SLICE_BLOCKS=2 HVR_KEY_VT_FOLD=0 HVR_NMS_MASK=1 HVR_PC_APPLY=0 HVR_READOUT_STREAMS=0 HVR_RPN_SIDE=0 HVR_RPN_WIDE=0 HVR_RPN_WIDE=16 HVR_SPLIT_NORMALIZE=0 HVR_SPLIT_NORMALIZE=1 HVR_STAGING=0"
for k in default $knobs; do
  if [ $k = default ]; then e=""; else e=$k; fi
  a=$(env $e timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -1 | cut -c1-90)
  b=$(env $e timeout 300 python bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-f32-leg --no-train-step --no-side-loops 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench ok %.1f' % d['value'])" 2>&1 | tail -1 | cut -c1-60)
  echo "$k | $a | $b"
done
