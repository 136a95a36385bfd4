cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for u in 0 -1; do
  export PGP_UNROLL=$u
  rm -rf /tmp/pq_$u
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/pq_$u -- python3 $R/tools/step_time.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pq_$u/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "score_hypotheses" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("score_hypotheses")[1][:12]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("unroll $u", k, {c: round(sum(v)/len(v)/1e6, 2) for c, v in d.items()}, "M per launch")
PY
done
