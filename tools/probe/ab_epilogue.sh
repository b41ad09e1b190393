P=tools/probe/libhvr_prev.so
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_s3.log 2>&1; tail -2 gpurun_out/pytest_s3.log
for i in 1 2; do
  echo "== prev bf16"; HVR_BENCH_LIB=$P python tools/probe/conv_hint_sweep.py --dtype bf16 --hints "" 2>/dev/null | cut -c1-60
  echo "== new bf16"; python tools/probe/conv_hint_sweep.py --dtype bf16 --hints "" 2>/dev/null | cut -c1-60
done
echo "== prev f16x2"; HVR_BENCH_LIB=$P python tools/probe/conv_hint_sweep.py --dtype f16x2 --hints "" 2>/dev/null | cut -c1-60
echo "== new f16x2"; python tools/probe/conv_hint_sweep.py --dtype f16x2 --hints "" 2>/dev/null | cut -c1-60
for i in 1 2; do
  echo "== prev"; HVR_BENCH_LIB=$P python tools/mode_window.py --mode f16x2 --iters 5 2>/dev/null | tail -1
  echo "== new"; python tools/mode_window.py --mode f16x2 --iters 5 2>/dev/null | tail -1
  echo "== prev"; HVR_BENCH_LIB=$P python tools/mode_window.py --mode bf16 --iters 8 2>/dev/null | tail -1
  echo "== new"; python tools/mode_window.py --mode bf16 --iters 8 2>/dev/null | tail -1
done
for i in 1 2; do
  echo "== prev bench"; HVR_BENCH_LIB=$P python tools/probe/bench_lib.py --no-cpu-baseline --no-train-step --no-f32-leg --no-side-loops 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['value_spread']['frames_per_s'], d['single_lane']['frames_per_s_per_gpu'], d['roofline']['frac'])"
  echo "== new bench"; python bench.py --no-cpu-baseline --no-train-step --no-f32-leg --no-side-loops 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['value_spread']['frames_per_s'], d['single_lane']['frames_per_s_per_gpu'], d['roofline']['frac'])"
done
