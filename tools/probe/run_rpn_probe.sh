cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "rpn or nms" > gpurun_out/rpn_tests.txt 2>&1
tail -5 gpurun_out/rpn_tests.txt
rm -rf /tmp/prof_rpn
HVR_RPN_WIDE=${WIDE:-1} timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_rpn -o rpn -- python tools/probe/rpn_probe.py > gpurun_out/rpn_prof_log.txt 2>&1
grep "us per call\|checksum" gpurun_out/rpn_prof_log.txt
python tools/rocpd_stats.py $(find /tmp/prof_rpn -name "*_results.db" | head -1) | grep -i "rpn\|nms\|kernel " > gpurun_out/rpn_prof_stats.txt
cat gpurun_out/rpn_prof_stats.txt
