export HVR_SPLIT_NORMALIZE=1
for t in 0 4 5 8 9 11 12; do echo -n "apply hint $t: "; HVR_TILE_APPLY_SPLIT=$t python tools/rel_bench.py --dtype f16x2 2>&1 | grep relation; done
for t in 0 1 4 5 8 9 12; do echo -n "scores hint $t: "; HVR_TILE_SCORES_SPLIT=$t python tools/rel_bench.py --dtype f16x2 2>&1 | grep relation; done
