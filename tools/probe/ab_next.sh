mkdir -p gpurun_out/r2
python -m pytest tests/test_kernels_gpu.py -x -q -k "tail" 2>&1 | tail -15
python -m pytest tests/test_parity_gpu.py -x -q -k "backbone or config1 or cached" 2>&1 | tail -5
for v in 1 0; do HVR_FUSE_NEXT=$v python bench.py --steps 20 --warmup 5 --no-side-loops --no-graphs --no-f32-leg --quick 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('FUSE_NEXT=$v', d['value'], d['ms_per_step'])
"; done
