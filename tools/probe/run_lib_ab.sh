cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "# bench.py headline (clip mode, two graph lanes) + eager single lane + relation core, two builds alternating on ONE box:"
echo "# pre = library 63b97b76 (commit e3cb963: before the ticket tail in gemm_tile.h), final = library 985b4476"
for rep in 1 2 3; do for lib in pre final; do
  if [ $lib = pre ]; then export HVR_BENCH_LIB=tools/probe/libhvr_pre_ticket.so; else unset HVR_BENCH_LIB; fi
  echo -n "$lib: "; timeout 300 python tools/probe/bench_with_lib.py --steps 20 --warmup 3 --repeats 3 --no-f32-leg --no-cpu-baseline --no-train-step --no-side-loops --no-graphs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('single_lane(eager headline)', d['value'], d['value_spread']['frames_per_s'], 'relation avg_ms', d['roofline']['avg_ms'], 'frac', d['roofline']['frac'])"
done; done
for rep in 1 2; do for lib in pre final; do
  if [ $lib = pre ]; then export HVR_BENCH_LIB=tools/probe/libhvr_pre_ticket.so; else unset HVR_BENCH_LIB; fi
  echo -n "$lib two lanes: "; timeout 300 python tools/probe/bench_with_lib.py --steps 20 --warmup 3 --repeats 3 --no-f32-leg --no-cpu-baseline --no-train-step --no-side-loops 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print(d['value'], d['value_spread']['frames_per_s'], 'single', d['single_lane']['frames_per_s_per_gpu'])"
done; done
} > gpurun_out/lib_ab.txt 2>&1
cat gpurun_out/lib_ab.txt
