#!/bin/bash
# tools/collect_cand8_pmc.sh <tag> -- the candidate FORMAT experiment priced with counters (VERDICT r4 task 4): the weighted
# scoring kernel with 16-byte float4 candidates (the product) and with 8-byte candidates (tools/ab/libpgp_cand8.so: fp16
# offsets from the cell centre + 16-bit ids, csrc/lcp_score.hip PGP_CAND8) at cell edges 0.85 / 0.7 / 0.6 delta.
# Per configuration: kernel duration (--kernel-trace --stats) and, in separate passes, SQ_INSTS_VALU / SALU / VMEM_RD / LDS,
# FETCH_SIZE, WRITE_SIZE, TCC hit / miss.  Output: gpurun_out/<tag>/cand8_pmc.json
set -u
TAG=${1:-cand8pmc}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--mode weighted --steps 20 --warmup 3 --no-cpu-baseline"
for CFG in float4:0.85 cand8:0.85 cand8:0.7 cand8:0.6 float4:0.7; do
  FMT=${CFG%%:*}; R=${CFG##*:}
  export PGP_CELL_RATIO=$R
  if [ $FMT = cand8 ]; then export PGP_LIB=$REPO/tools/ab/libpgp_cand8.so; else unset PGP_LIB; fi
  K=${FMT}_$R
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$K -- python3 $REPO/bench.py $ARGS > $OUT/bench_$K.json 2> $OUT/trace_$K.err
  cp $(find $OUT/trace_$K -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$K.csv 2>/dev/null
  cp $REPO/gpurun_out/bench_detail_n1.json $OUT/detail_$K.json 2>/dev/null
  i=0
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${K}_$i -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_${K}_$i.err || echo "pass $K $i failed" >> $OUT/failed.txt
  done
  echo "done $K"
done
unset PGP_CELL_RATIO PGP_LIB
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for K in ("float4_0.85", "cand8_0.85", "cand8_0.7", "cand8_0.6", "float4_0.7"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/pmc_{K}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "score_hypotheses_flat<1>" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    row = {c: sum(v) / len(v) for c, v in agg.items()}
    try:
        for r in csv.DictReader(open(f"{out}/kernel_stats_{K}.csv")):
            if "score_hypotheses_flat<1>" in r["Name"]:
                row["kernel_avg_us"] = float(r["AverageNs"]) / 1e3
                row["calls"] = int(r["Calls"])
    except Exception as e:
        row["kernel_stats_error"] = repr(e)
    try:
        b = json.load(open(f"{out}/detail_{K}.json"))
        row["ms_per_step"] = b["ms_per_step"]
        row["index"] = {k: b["index"][k] for k in ("cell_size", "n_candidates", "bytes_index", "n_occupied")}
    except Exception as e:
        row["bench_error"] = repr(e)
    if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
        row["hbm_MB_per_launch"] = (2.0 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024 / 1e6   # gfx950: FETCH_SIZE reports half of wide reads
    res[K] = row
json.dump({"how": "tools/collect_cand8_pmc.sh: weighted scoring kernel of `bench.py --mode weighted --steps 20`, candidate format x PGP_CELL_RATIO; per-launch "
                  "averages; FETCH_SIZE / WRITE_SIZE in KB; cand8 = tools/ab/libpgp_cand8.so (PGP_CAND8 build: results approximate, timing experiment)",
           "configs": res}, open(out + "/cand8_pmc.json", "w"), indent=1)
for K, row in res.items():
    print(K, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in row.items()})
PY
