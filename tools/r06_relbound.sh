#!/bin/bash
# Upper bounds for the grouped relation core (four windows per call): timing-only builds that delete one ingredient of a kernel
# (results are NOT valid outputs) -- what a perfect overlap of that ingredient could buy at most.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_relbound; mkdir -p $O
for lib in product btnoepi btnodma btnomma btnoreads abnoadj; do
  if [ $lib = product ]; then unset HVR_BENCH_LIB; else export HVR_BENCH_LIB=abtest/libhvr_$lib.so; fi
  rm -rf /tmp/rb; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/rb -o rel -- python tools/rel_bench.py --iters 10 --groups 4 > /tmp/rb.log 2>&1
  echo "== $lib" >> $O/relation_bounds.txt
  python tools/rocpd_stats.py $(find /tmp/rb -name "*.db" | head -1) 2>/dev/null | grep -E "relation_(scores|apply)_bt" | cut -c1-60,110-170 >> $O/relation_bounds.txt
done
cat $O/relation_bounds.txt
