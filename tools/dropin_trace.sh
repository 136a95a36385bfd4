#!/bin/bash
# kernel trace of 50 in-memory drop-in calls (shim/test_shim): which kernels a call launches and how long they run
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/dropin_trace
mkdir -p $OUT
python3 - "$OUT" <<'PY'
import os, sys, tempfile, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from _dropin import make_dropin_case
d = os.path.join(sys.argv[1], "case"); os.makedirs(d, exist_ok=True)
args, case = make_dropin_case(d)
open(os.path.join(sys.argv[1], "args.txt"), "w").write("\n".join(args))
PY
mapfile -t ARGS < $OUT/args.txt
rm -rf $OUT/trace   # (an earlier run's summary must not be picked up below)
cd /tmp && export TMPDIR=/tmp
PGP_SHIM_SEED=12345 SHIM_TEST_REPEAT=50 SHIM_TEST_INMEMORY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $REPO/shim/test_shim "${ARGS[@]}" > $OUT/out.txt 2> $OUT/err.txt
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python3 - "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/kernel_stats.csv")))
tot = 0.0
print(f"{'kernel':60s} {'calls/50':>9s} {'avg us':>8s} {'us per drop-in call':>20s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    name = r["Name"].replace("pgp::(anonymous namespace)::", "").replace("void ", "")
    name = name[:name.find("(")] if "(" in name else name
    per = float(r["TotalDurationNs"]) / 50 / 1e3
    tot += per
    if per >= 1.0: print(f"{name[:60]:60s} {int(r['Calls'])/50:9.1f} {float(r['AverageNs'])/1e3:8.1f} {per:20.1f}")
print(f"kernels of one drop-in call, summed: {tot:.0f} us in {sum(int(r['Calls']) for r in rows)/50:.0f} launches")
PY
