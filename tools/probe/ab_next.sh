A="--no-cpu-baseline --no-train-step --no-f32-leg --no-side-loops --steps 40 --warmup 5"
pick='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["single_lane"]["frames_per_s_per_gpu"])'
for i in 1 2 3; do
echo -n "big expand: "; python bench.py $A 2>/dev/null | python -c "$pick"
echo -n "panel expand: "; HVR_BIGTILE_RES=0 python bench.py $A 2>/dev/null | python -c "$pick"
done
echo -n "no big tiles at all: "; HVR_BIGTILE=0 python bench.py $A 2>/dev/null | python -c "$pick"
