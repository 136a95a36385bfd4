#!/bin/bash
# time line of ONE in-memory drop-in call (the 30th of 50): every kernel / copy with its queue, start offset, duration and the
# gap since the previous end on the same queue -- where the call's critical path runs
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/dropin_timeline
mkdir -p $OUT
python3 - "$OUT" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from _dropin import make_dropin_case
d = os.path.join(sys.argv[1], "case"); os.makedirs(d, exist_ok=True)
args, case = make_dropin_case(d)
open(os.path.join(sys.argv[1], "args.txt"), "w").write("\n".join(args))
PY
mapfile -t ARGS < $OUT/args.txt
rm -rf $OUT/trace
cd /tmp && export TMPDIR=/tmp
export PGP_SHIM_SEED=12345 SHIM_TEST_REPEAT=50 ${TIMELINE_ENV:-SHIM_TEST_INMEMORY=1}
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- $REPO/shim/test_shim "${ARGS[@]}" > $OUT/out.txt 2> $OUT/err.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("pgp::(anonymous namespace)::", "").replace("void ", "")
        name = name[:name.find("(")] if "(" in name else name
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), name[:46]))
for f in glob.glob(out + "/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "copy")[:46]))
ev.sort()
# calls are separated by the longest idle gaps: find the starts of select_bases (one per call)
starts = [i for i, e in enumerate(ev) if e[3].startswith("select_bases")]
if len(starts) < 32:
    print("calls found:", len(starts)); sys.exit(0)
# a call begins with its first event after the previous call's last (registered_points...): take events between the 30th
# select_bases's predecessors: walk back to the largest gap before it
def call_span(k):
    i = starts[k]
    lo = i
    while lo > 0 and ev[lo][0] - max(e[1] for e in ev[max(0, lo - 8):lo]) < 40000 and lo > starts[k - 1]:
        lo -= 1
    hi = starts[k + 1]
    while hi > i and ev[hi][0] - max(e[1] for e in ev[max(0, hi - 8):hi]) < 40000 and hi > i:
        hi -= 1
    return lo, hi
lo, hi = call_span(30)
t0 = ev[lo][0]
last_end = {}
print(f"{'t us':>8s} {'dur':>7s} {'gap(q)':>7s} queue  what")
for s, e, q, name in ev[lo:hi]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {gap:7.1f} {q:6s} {name}")
print(f"span {(max(e[1] for e in ev[lo:hi]) - t0) / 1e3:.1f} us, {hi - lo} events")
PY
