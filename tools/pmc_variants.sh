#!/bin/bash
# tools/pmc_variants.sh -- counters for kernel variants (PGP_UNROLL) in separate rocprofv3 passes
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  export PGP_UNROLL=$V
  OUT=$REPO/gpurun_out/pmcv_$V; mkdir -p $OUT
  i=0
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/p$i.err || echo "pass $i failed" >> $OUT/failed.txt
  done
done
python3 - $REPO/gpurun_out "$@" <<'PY'
import csv, glob, sys, collections, json
root = sys.argv[1]
for v in sys.argv[2:]:
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{root}/pmcv_{v}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "score_hypotheses" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("variant", v, json.dumps({k: round(sum(x) / len(x)) for k, x in sorted(agg.items())}))
PY
