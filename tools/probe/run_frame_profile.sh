cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/f_ks; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/f_ks -o fr -- python tools/probe/frame_profile.py 20 > gpurun_out/frame_profile.log 2>&1
grep "one frame" gpurun_out/frame_profile.log
python tools/rocpd_stats.py $(find /tmp/f_ks -name "*.db" | head -1) > gpurun_out/frame_kernel_stats.txt
head -32 gpurun_out/frame_kernel_stats.txt | cut -c1-175
