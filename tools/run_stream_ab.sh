cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 0 4 0 4; do echo "== HVR_RPN_WIDE=$w"; HVR_RPN_WIDE=$w timeout 300 python tools/stream_bench.py --steps 60 2>&1 | grep -v amdgpu.ids | tail -1; done > gpurun_out/stream_ab.txt 2>&1
cat gpurun_out/stream_ab.txt
