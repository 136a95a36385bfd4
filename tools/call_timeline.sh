#!/bin/bash
# tools/call_timeline.sh <row>: rocprofv3 kernel + copy trace of tools/call_timeline.py, the operations of the call between the two idle gaps
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
ROW=${1:-voxel}
OUT=$REPO/gpurun_out/call_timeline_$ROW
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $REPO/tools/call_timeline.py $ROW > $OUT/out.txt 2> $OUT/err.txt
grep CALL_MS $OUT/out.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("pgp::(anonymous namespace)::", "").replace("void ", "")
        name = name[:name.find("(")] if "(" in name else name
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:50]))
for f in glob.glob(out + "/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "copy")[:50]))
ev.sort()
# the call that is looked at lies between the two longest idle gaps (>= 15 ms) of the run's tail
gaps = [(ev[i + 1][0] - ev[i][1], i) for i in range(len(ev) - 1)]
big = sorted(i for g, i in gaps if g > 15e6)
if len(big) < 2:
    print("no isolated call found"); sys.exit(0)
lo, hi = big[-2] + 1, big[-1] + 1
t0, prev = ev[lo][0], ev[lo][0]
for s, e, name in ev[lo:hi]:
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {name}")
    prev = e
print(f"span {(ev[hi - 1][1] - t0) / 1e3:.1f} us, {hi - lo} operations, {sum(e - s for s, e, _ in ev[lo:hi]) / 1e3:.1f} us of them busy")
PY
