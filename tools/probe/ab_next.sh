A="--no-cpu-baseline --no-train-step --no-f32-leg --no-side-loops --steps 30 --warmup 5"
pick='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["single_lane"]["frames_per_s_per_gpu"], d["graphed_stream"]["frames_per_s_per_gpu"])'
for i in 1 2; do
echo -n "base: "; python tools/probe/bench_lib.py $A 2>/dev/null | python -c "$pick"
echo -n "nt:   "; HVR_BENCH_LIB=dbg/libhvr_nt.so python tools/probe/bench_lib.py $A 2>/dev/null | python -c "$pick"
done
