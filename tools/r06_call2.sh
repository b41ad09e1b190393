#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c2; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/lmp tools/lane_map_probe.hip > /dev/null 2>&1 && /tmp/lmp > $O/lane_map_probe.txt 2>&1
cat $O/lane_map_probe.txt
for m in f16x2 f32; do
  timeout 600 python tools/noise_contrib.py --mode $m --head selsa --clip 0 > $O/noise_contrib_${m}_selsa.txt 2>&1
  tail -9 $O/noise_contrib_${m}_selsa.txt
done
timeout 600 python tools/noise_contrib.py --mode f16x2 --head hvr --clip 7 > $O/noise_contrib_f16x2_hvr_clip7.txt 2>&1
tail -9 $O/noise_contrib_f16x2_hvr_clip7.txt
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log )
tail -4 $O/pytest.log
