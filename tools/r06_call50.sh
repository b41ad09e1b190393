#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r06_final_5.log 2>&1; tail -3 gpurun_out/gputest_r06_final_5.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
ROUND=r06 bash tools/collect_profiles.sh A > /dev/null 2>&1
ROUND=r06 bash tools/collect_profiles.sh C > /dev/null 2>&1
ls gpurun_out/profiles | wc -l
