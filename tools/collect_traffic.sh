#!/bin/bash
# The short pass of tools/collect_profiles.sh that must be repeated whenever libhvr_hip.so changes: the relation core's HBM-side
# bytes per launch (FETCH_SIZE / WRITE_SIZE, separate PMC runs, kernel-trace only) -> gpurun_out/profiles/relation_traffic.json,
# keyed to the library's sha256 (bench.py reports `roofline.traffic` only for the build the JSON names).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/r_$c; rocprofv3 --kernel-trace --pmc $c -d /tmp/r_$c -o rel -- python tools/rel_bench.py --iters 5 > /dev/null 2>&1
  python tools/rocpd_pmc.py $(find /tmp/r_$c -name "*.db" | head -1) $c > $out/rel_pmc_$(echo $c | tr A-Z a-z).txt
done
python tools/make_traffic_json.py $(find /tmp/r_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/r_WRITE_SIZE -name "*.db" | head -1) $out/relation_traffic.json
cat $out/relation_traffic.json
