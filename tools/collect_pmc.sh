#!/bin/bash
# tools/collect_pmc.sh <tag> [bench args...] -- rocprofv3 passes for the scoring kernel on the GPU box.
# One --kernel-trace --stats pass, then separate --pmc passes (TCC slots: FETCH_SIZE=3, WRITE_SIZE=2).
# Run via gpurun from the repo root; outputs under gpurun_out/<tag>/.
set -u
TAG=${1:-pmc}; shift || true
ARGS=${@:-"--steps 20 --warmup 3 --no-cpu-baseline"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVE32_INSTS" \
         "FETCH_SIZE" "WRITE_SIZE" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc$i.err || echo "pass $i ($C) failed" >> $OUT/failed.txt
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
import re
for k, d in agg.items():
    name = k.replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+(?:<[^>]*>)?)\(", name)
    short = m.group(1) if m else name
    if any(t in short for t in ("score_hypotheses", "icp_refine", "finalize_scores", "q_match", "pair_rows")):
        res[short] = {c: sum(v) / len(v) for c, v in d.items()} | {"n": max(len(v) for v in d.values())}
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
