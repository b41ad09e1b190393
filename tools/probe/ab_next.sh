A="--no-cpu-baseline --no-train-step --no-f32-leg --no-side-loops --no-graphs --lanes 1 --steps 30 --warmup 5"
pick='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); k=d["kernel_classes"]; print(d["value"], round(k["conv"]["ms"]+k["conv_expand"]["ms"],3))'
for i in 1 2 3; do
echo -n "fused: "; python bench.py $A 2>/dev/null | python -c "$pick"
echo -n "split: "; HVR_TAIL_TILE=0 python bench.py $A 2>/dev/null | python -c "$pick"
done
