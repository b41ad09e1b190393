cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_s9.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest_s9.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_s9.log 2>&1; tail -2 gpurun_out/smoke_s9.log
rm -rf /tmp/prof_rpn
HVR_RPN_WIDE=4 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_rpn -o rpn -- python tools/probe/rpn_probe.py > gpurun_out/rpn_wide_probe.txt 2>&1
python tools/rocpd_stats.py $(find /tmp/prof_rpn -name "*_results.db" | head -1) | grep -i "rpn\|nms\|kernel \|dispatches" > gpurun_out/rpn_wide_kernel_stats.txt
timeout 300 python tools/stream_bench.py --steps 60 2>/dev/null | tail -1 > gpurun_out/stream_bench_s9.json
cat gpurun_out/stream_bench_s9.json
