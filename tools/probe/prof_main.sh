# usage: prof_main.sh <tag> [ENV=val ...]   kernel stats + window phases of the headline loop
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out/r2
python bench.py --steps 20 --warmup 5 --no-side-loops --no-graphs --no-f32-leg --quick 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag', d['value'], d['ms_per_step'])
"
rm -rf /tmp/p_ks; rocprofv3 --kernel-trace --stats -d /tmp/p_ks -o bench -- python bench.py --steps 5 --warmup 2 --no-side-loops --no-graphs --no-f32-leg --quick > gpurun_out/r2/prof_$tag.json 2>/dev/null
python tools/rocpd_stats.py $(find /tmp/p_ks -name "*.db" | head -1) > gpurun_out/r2/prof_${tag}_stats.txt
python tools/rocpd_phases.py $(find /tmp/p_ks -name "*.db" | head -1) 4 > gpurun_out/r2/prof_${tag}_phases.txt 2>&1
