cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
rm -rf /tmp/gs; rocprofv3 --kernel-trace --stats -d /tmp/gs -o gs -- python tools/probe/graph_stream.py 8 > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/gs -name "*.db" | head -1) > gpurun_out/r2b/stream_stats.txt
python tools/rocpd_phases.py $(find /tmp/gs -name "*.db" | head -1) 4 > gpurun_out/r2b/stream_phases.txt 2>&1
