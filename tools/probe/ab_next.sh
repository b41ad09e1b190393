for v in 0 9 8 4; do echo -n "HVR_TILE_SCORES=$v: "; HVR_TILE_SCORES=$v python tools/rel_bench.py --mq 300 --iters 50 2>&1 | grep relation; done
