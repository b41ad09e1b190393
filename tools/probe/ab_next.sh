for w in 256 384 512 768; do for k in 8 16 30; do for per in 2 4 6; do
echo -n "wgs $w mink $k per $per: "; HVR_CONV_SPLITK_WGS=$w HVR_CONV_SPLITK_MINK=$k HVR_CONV_SPLITK_PER=$per python tools/probe/stream_pipe.py 2>&1 | grep -v amdgpu | sed -n 5,8p | awk '{printf "%s ", $(NF-3)}'; echo
done; done; done
