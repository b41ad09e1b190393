cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "relation or key_stage" 2>&1 | tail -3
for m in 0 1 0 1; do HVR_KEY_MERGE=$m timeout 120 python tools/key_bench.py --iters 200 2>&1 | grep "us per call"; done
rm -rf /tmp/k_ks; HVR_KEY_MERGE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k_ks -o key -- python tools/key_bench.py --iters 50 > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/k_ks -name "*.db" | head -1) | head -8 | cut -c1-190
rm -rf /tmp/k_ks0; HVR_KEY_MERGE=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k_ks0 -o key -- python tools/key_bench.py --iters 50 > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/k_ks0 -name "*.db" | head -1) | head -8 | cut -c1-190
} > gpurun_out/key_stage_ab.txt 2>&1
cat gpurun_out/key_stage_ab.txt
