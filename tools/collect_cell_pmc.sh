#!/bin/bash
# tools/collect_cell_pmc.sh <tag> -- "fewer candidates per query" priced with counters: the weighted scoring kernel
# at cell edges 0.85 (default), 0.7, 0.6, 0.5 and 0.425 delta (PGP_CELL_RATIO; 0.425 = the candidate lists an
# octant split of every cell would give).  Per ratio: kernel duration (--kernel-trace --stats) and, in separate
# passes, SQ_INSTS_VALU / SQ_WAVES, FETCH_SIZE, WRITE_SIZE, TCC hit / miss.  Output: gpurun_out/<tag>/cell_pmc.json
set -u
TAG=${1:-cellpmc}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--mode weighted --steps 20 --warmup 3 --no-cpu-baseline"
for R in 0.85 0.7 0.6 0.5 0.425; do
  export PGP_CELL_RATIO=$R
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$R -- python3 $REPO/bench.py $ARGS > $OUT/bench_$R.json 2> $OUT/trace_$R.err
  cp $(find $OUT/trace_$R -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$R.csv 2>/dev/null
  i=0
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${R}_$i -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_${R}_$i.err || echo "pass $R $i failed" >> $OUT/failed.txt
  done
done
unset PGP_CELL_RATIO
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json, re
out = sys.argv[1]
res = {}
for R in ("0.85", "0.7", "0.6", "0.5", "0.425"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/pmc_{R}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "score_hypotheses_flat<1>" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    row = {c: sum(v) / len(v) for c, v in agg.items()}
    try:
        for r in csv.DictReader(open(f"{out}/kernel_stats_{R}.csv")):
            if "score_hypotheses_flat<1>" in r["Name"]:
                row["kernel_avg_us"] = float(r["AverageNs"]) / 1e3
                row["calls"] = int(r["Calls"])
    except Exception as e:
        row["kernel_stats_error"] = repr(e)
    try:
        b = json.loads(open(f"{out}/bench_{R}.json").read().strip().splitlines()[-1])
        row["ms_per_step"] = b["ms_per_step"]
        row["index"] = {k: b["index"][k] for k in ("cell_size", "n_candidates", "bytes_index", "n_occupied")}
    except Exception as e:
        row["bench_error"] = repr(e)
    res[R] = row
json.dump({"how": "tools/collect_cell_pmc.sh: weighted scoring kernel of `bench.py --mode weighted --steps 20` per PGP_CELL_RATIO; per-launch averages; FETCH_SIZE / WRITE_SIZE in KB", "ratios": res}, open(out + "/cell_pmc.json", "w"), indent=1)
for R, row in res.items():
    print(R, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in row.items()})
PY
