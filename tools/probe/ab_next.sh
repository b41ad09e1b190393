A="--no-cpu-baseline --no-train-step --no-f32-leg --no-side-loops --steps 40 --warmup 5"
pick='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["single_lane"]["frames_per_s_per_gpu"])'
run() { echo -n "$1: "; env $2 python bench.py $A 2>/dev/null | python -c "$pick"; }
for i in 1 2; do
run "default            " "X=1"
run "expand nc 4        " "HVR_EXPAND_NC=4"
run "frame groups 2     " "HVR_FRAME_GROUPS=2"
run "readout 1 stream   " "HVR_READOUT_STREAMS=0"
run "gm_apply 4         " "HVR_GM_APPLY=4"
run "gm_scores 4        " "HVR_GM_SCORES=4"
run "pc apply           " "HVR_PC_APPLY=1"
done
