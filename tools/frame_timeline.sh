#!/bin/bash
# the device's view of ONE frame of n objects through getProbableTransformsSuper4PCSFrame (the 20th of 30): per hardware queue the busy
# time and what ran, the union of all queues' busy time against the frame's span
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/frame_timeline
N=${1:-3}
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
export PGP_SHIM_SEED=12345 SHIM_TEST_REPEAT=30 SHIM_TEST_FRAME=$N PGP_SHIM_PRIVATE_RAND=1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- $REPO/shim/test_shim "${ARGS[@]}" > $OUT/out.txt 2> $OUT/err.txt
python3 - "$OUT" "$N" <<'PY'
import csv, glob, sys
out, n = sys.argv[1], int(sys.argv[2])
ev = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("pgp::(anonymous namespace)::", "").replace("void ", "")
        name = name[:name.find("(")] if "(" in name else name
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), name[:40]))
for f in glob.glob(out + "/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "copy")[:40]))
ev.sort()
sel = [i for i, e in enumerate(ev) if e[3].startswith("select_bases")]
# frames: groups of n select_bases launches (after the single call that precedes the frames)
first = 20 * n
if len(sel) < first + 2 * n:
    print("select_bases launches found:", len(sel)); sys.exit(0)
lo_t = ev[sel[first]][0] - 150000          # (the frame's uploads start ~0.1 ms before its first select_bases)
hi_t = ev[sel[first + n]][0] - 150000
fr = [e for e in ev if lo_t <= e[0] < hi_t]
t0 = min(e[0] for e in fr)
span = max(e[1] for e in fr) - t0
iv = sorted((e[0], e[1]) for e in fr)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"frame of {n}: {len(fr)} device operations, span {span / 1e3:.1f} us, at least one queue busy {busy / 1e3:.1f} us ({100.0 * busy / span:.0f} %), "
      f"sum of all operations {sum(e[1] - e[0] for e in fr) / 1e3:.1f} us")
qs = {}
for s, e, q, name in fr:
    qs.setdefault(q, []).append((s, e, name))
for q, l in sorted(qs.items()):
    tot = sum(e - s for s, e, _ in l)
    big = sorted(l, key=lambda x: x[0] - x[1])[:4]
    print(f"  {q:5s} {len(l):3d} operations, {tot / 1e3:7.1f} us busy, first at {(l[0][0] - t0) / 1e3:6.1f}, last ends {(l[-1][1] - t0) / 1e3:6.1f};  longest: "
          + ", ".join(f"{nm} {(e - s) / 1e3:.0f}" for s, e, nm in big))
PY
