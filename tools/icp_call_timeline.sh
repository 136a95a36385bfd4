#!/bin/bash
# time line of one host-pointer ICP call from near start: copies, kernels, gaps (where the call's fixed cost sits)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/icp_timeline
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in 1 64; do
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/p$P -- python3 $REPO/tools/icp_one.py $P 0.3 12 > $OUT/p$P.out 2> $OUT/p$P.err
python3 - "$OUT/p$P" $P <<'PY'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("pgp::(anonymous namespace)::", "").replace("void ", "")
        name = name[:name.find("(")] if "(" in name else name
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:60]))
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "copy")[:40]))
ev.sort()
# the last three calls: split at gaps > 60 us before an H2D copy
starts = [i for i, e in enumerate(ev) if e[2].startswith("MEMORY_COPY_HOST_TO_DEVICE")]
print("poses", sys.argv[2], "events", len(ev))
for k in starts[-3:]:
    t0 = ev[k][0]
    prev_end = ev[k - 1][1] if k else t0
    print(f"  -- call (idle before: {(t0 - prev_end) / 1e3:.1f} us)")
    i = k
    last = t0
    while i < len(ev) and (i == k or not ev[i][2].startswith("MEMORY_COPY_HOST_TO_DEVICE")):
        s, e, n = ev[i]
        print(f"  {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - last) / 1e3:6.1f}  {n}")
        last = e
        i += 1
PY
done
