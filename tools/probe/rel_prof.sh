# usage: rel_prof.sh <lib or ""> [rel_bench args]   per-kernel times of the relation core
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
lib=$1; shift
[ -n "$lib" ] && export HVR_BENCH_LIB=$lib
rm -rf /tmp/p_rel; rocprofv3 --kernel-trace --stats -d /tmp/p_rel -o rel -- python tools/rel_bench.py --iters 20 "$@" > /dev/null 2>&1
echo "== lib=${lib:-shipped} $@"
python tools/rocpd_stats.py $(find /tmp/p_rel -name "*.db" | head -1) | grep -i "relation\|tile_kernel\|pc_tile" | cut -c1-60,100-170
