#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/icp_table
mkdir -p $OUT; rm -rf $OUT/trace
python3 $REPO/tools/icp_table.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/icp_table.py > $OUT/out.txt 2> $OUT/err.txt
python3 - "$(find $OUT/trace -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:10]:
    name = r["Name"].replace("pgp::(anonymous namespace)::", "").replace("void ", "")
    name = name[:name.find("(")] if "(" in name else name
    print(f"{name[:50]:50s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
