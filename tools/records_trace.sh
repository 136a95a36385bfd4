#!/bin/bash
# per-kernel time of a weighted C2 step with pgp_set_exact_records on (tools/records_trace.py under rocprofv3 --kernel-trace --stats)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/records_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/records_trace.py > $OUT/out.txt 2> $OUT/err.txt
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python3 - "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/kernel_stats.csv")))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    name = r["Name"].replace("pgp::(anonymous namespace)::", "").replace("void ", "")
    name = name[:name.find("(")] if "(" in name else name
    print(f"{name[:60]:60s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
