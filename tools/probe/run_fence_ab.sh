cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "# key stage 300 x 4500 x 1024 bf16, tools/probe/key_bench.py --iters 200; libs from tools/build_dbg.sh (withfence = product source, nofence = -DHVR_DBG_MERGE_NOFENCE: timing only)"
for rep in 1 2; do
  for cfg in "withfence 0" "withfence 1" "nofence 1"; do set -- $cfg
    echo -n "$1 HVR_KEY_MERGE=$2: "; HVR_BENCH_LIB=tools/probe/libhvr_$1.so HVR_KEY_MERGE=$2 timeout 120 python tools/probe/key_bench.py --iters 200 --repeat 50 --dump /tmp/key_$1_$2.npy 2>&1 | grep "us per call\|repeat\|Error\|error" | tr '\n' ' '; echo
  done
done
python - <<'P'
import numpy as np
ref = np.load('/tmp/key_withfence_0.npy')
for n in ('withfence_1', 'nofence_1'):
    a = np.load('/tmp/key_%s.npy' % n)
    print(n, 'equals the reduce launch bit for bit:', bool(np.array_equal(a, ref)), ' differing elements:', int((a != ref).sum()))
P
for cfg in "withfence 1" "nofence 1"; do set -- $cfg
  rm -rf /tmp/k_$1; HVR_BENCH_LIB=tools/probe/libhvr_$1.so HVR_KEY_MERGE=$2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k_$1 -o key -- python tools/probe/key_bench.py --iters 50 > /dev/null 2>&1
  echo "== kernel stats, $1 HVR_KEY_MERGE=$2"; python tools/rocpd_stats.py $(find /tmp/k_$1 -name "*.db" | head -1) | head -6 | cut -c1-185
done
} > gpurun_out/key_fence_ab.txt 2>&1
cat gpurun_out/key_fence_ab.txt
