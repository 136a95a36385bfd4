#!/bin/bash
# tools/collect_rows_pmc.sh <tag> -- kernel-trace stats + two PMC passes over tools/profile_rows.py
# (every kernel family besides the scoring kernel).  Outputs under gpurun_out/<tag>/.
set -u
TAG=${1:-rows}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/profile_rows.py > $OUT/trace.out 2> $OUT/trace.err
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_rows.csv 2>/dev/null
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  PGP_PROFILE_REPS=2 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 $REPO/tools/profile_rows.py > /dev/null 2> $OUT/pmc_$i.err || echo "pass $i ($C) failed" >> $OUT/failed.txt
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    name = k.replace("(anonymous namespace)::", "").replace("pgp::", "")
    m = re.search(r"(\w+(?:<[^>]*>)?)\(", name)
    short = m.group(1) if m else name[:60]
    if short.startswith(("__amd", "void at::", "at::")) or "rocprim" in name:
        continue
    res[short] = {c: sum(v) / len(v) for c, v in d.items()} | {"n": max(len(v) for v in d.values())}
json.dump({"how": "tools/collect_rows_pmc.sh: rocprofv3 --pmc <group> --kernel-trace over tools/profile_rows.py, per-launch averages; FETCH_SIZE / WRITE_SIZE in KB", "kernels": res}, open(out + "/pmc_rows.json", "w"), indent=1)
print(sorted(res))
PY
rm -rf $OUT/pmc_[0-9] $OUT/trace
