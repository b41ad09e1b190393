mkdir -p gpurun_out/r2b
for i in 1 2 3; do python tools/rel_bench.py --iters 20; HVR_BENCH_LIB=dbg/libhvr_head.so python tools/rel_bench.py --iters 20; done > gpurun_out/r2b/rel_ab.txt 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2b/tests.txt
