#!/bin/bash
# LDS counters of the persistent ICP kernel on the far-start problem: is phase B bound by the LDS array?
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/icp_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_LDS_MEM_VIOLATIONS" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  for P in 64 256; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p${P}_$i -- python3 $REPO/tools/icp_one.py $P 5 3 > /dev/null 2> $OUT/p${P}_$i.err || echo "pass $i poses $P ($C) failed" >> $OUT/failed.txt
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for P in (64, 256):
    agg = collections.defaultdict(list)
    for f in glob.glob(out + f"/p{P}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "icp_persist" in r["Kernel_Name"] and "CLUSTER" not in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[P] = {c: sum(v) / len(v) for c, v in agg.items()}
    print(P, json.dumps(res[P]))
json.dump(res, open(out + "/icp_lds.json", "w"), indent=1)
PY
