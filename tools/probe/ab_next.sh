export HVR_LANES_LIST=2
for i in 1 2 3; do
echo -n "thr hint, min 96:  "; HVR_THR=1 python tools/probe/graph_lanes.py 60 2>&1 | grep lanes
echo -n "thr hint, min 128: "; HVR_THR=1 HVR_BIGTILE_MIN_SHARED=128 python tools/probe/graph_lanes.py 60 2>&1 | grep lanes
echo -n "no hint:           "; HVR_THR=0 python tools/probe/graph_lanes.py 60 2>&1 | grep lanes
done
