#!/bin/bash
# timing experiment: icp_refine<true> (one iteration's update, host-driven ICP) with parts cut out (tools/ab/libpgp_refabl<k>.so,
# wrong results), on the reference's table-alignment shape -- where its ~88 us per launch go
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for k in 0 1 2 3 4; do
  if [ $k = 0 ]; then lib=""; else lib=$REPO/tools/ab/libpgp_refabl$k.so; fi
  OUT=$REPO/gpurun_out/refabl_$k; mkdir -p $OUT; rm -rf $OUT/trace
  ( cd /tmp && export TMPDIR=/tmp && PGP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/icp_table.py > $OUT/out.txt 2> $OUT/err.txt )
  python3 - "$(find $OUT/trace -name '*kernel_stats.csv' | head -1)" $k <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "icp_refine" in r["Name"]:
        print(f"ablation {sys.argv[2]}: icp_refine<true> avg {float(r['AverageNs'])/1e3:6.1f} us over {r['Calls']} launches")
PY
done
