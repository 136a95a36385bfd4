#!/bin/bash
# tools/collect_pmc.sh <tag> -- rocprofv3 passes for the scoring kernel on the GPU box, both modes.
# Per mode: one --kernel-trace --stats pass, then separate --pmc passes (TCC slots: FETCH_SIZE=3,
# WRITE_SIZE=2; never combined with trace domains other than the kernel trace).  Run via gpurun from
# the repo root; outputs under gpurun_out/<tag>/:
#   pmc_current.json   per-launch averages of every counter, per kernel, + the kernel source id
#                      (copy to profiles/pmc_current.json: bench.py's roofline reads it)
#   kernel_stats_<mode>.csv   the --stats summary of `python3 bench.py --mode <mode> ...`
set -u
TAG=${1:-pmc}
ARGS="--steps 20 --warmup 3 --no-cpu-baseline"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in weighted plain; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$MODE -- python3 $REPO/bench.py --mode $MODE $ARGS > $OUT/bench_$MODE.json 2> $OUT/trace_$MODE.err
  cp $(find $OUT/trace_$MODE -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$MODE.csv 2>/dev/null
  i=0
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${MODE}_$i -- python3 $REPO/bench.py --mode $MODE $ARGS > /dev/null 2> $OUT/pmc_${MODE}_$i.err || echo "pass $MODE $i ($C) failed" >> $OUT/failed.txt
  done
done
# calibration of the L1 counter unit on a known access pattern (tools/peaks.hip: 64 / 16 / 4 lines per load)
if [ -x $REPO/tools/ab/peaks ]; then cp $REPO/tools/ab/peaks /tmp/peaks; fi
if [ -x /tmp/peaks ]; then
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_peaks -- /tmp/peaks > $OUT/peaks_under_pmc.json 2> $OUT/pmc_peaks.err
fi
# static VALU mix of the kernels as built (tools/valu_mix.py prices the VALU unit with it)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math --cuda-device-only \
  -S $REPO/physimglobalpose_amd/csrc/lcp_score.hip -o $OUT/lcp_score.s 2> $OUT/lcp_score_s.err
python3 - "$OUT" "$REPO" <<'PY'
import csv, glob, sys, collections, json, re, hashlib, os, datetime
out, repo = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(repo, "tools"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    name = k.replace("(anonymous namespace)::", "").replace("pgp::", "")
    m = re.search(r"(\w+(?:<[^>]*>)?)\(", name)
    short = m.group(1) if m else name
    if any(t in short for t in ("score_hypotheses", "finalize_scores", "valu_kernel", "l1_kernel")):
        res[short] = {c: sum(v) / len(v) for c, v in d.items()} | {"n": max(len(v) for v in d.values())}
h = hashlib.sha256()
for f in ("lcp_score.hip", "grid_index.hip", "pgp_internal.h"):
    h.update(open(os.path.join(repo, "physimglobalpose_amd", "csrc", f), "rb").read())
for line in open(os.path.join(repo, "physimglobalpose_amd", "csrc", "Makefile")):   # same rule as bench.py kernel_source_id()
    if line.startswith("FLAGS") or line.startswith("ARCH"):
        h.update(line.strip().encode())
try:
    import valu_mix
    text = open(out + "/lcp_score.s").read()
    mix = {"score_hypotheses_flat<0>": valu_mix.kernel_mix(text, "score_hypotheses_flatILi0E"),
           "score_hypotheses_flat<1>": valu_mix.kernel_mix(text, "score_hypotheses_flatILi1E")}
except Exception as e:   # the counters are still worth keeping
    mix = {"error": repr(e)}
doc = {"source_id": h.hexdigest()[:16], "collected": datetime.date.today().isoformat(), "valu_mix": mix,
       "how": "tools/collect_pmc.sh: rocprofv3 --pmc <group> --kernel-trace, one group per pass, over "
              "`python3 bench.py --mode <mode> --steps 20 --warmup 3 --no-cpu-baseline` (C2, 8 distinct batches "
              "in rotation); values are averages per launch; FETCH_SIZE / WRITE_SIZE in KB",
       "kernels": res}
json.dump(doc, open(out + "/pmc_current.json", "w"), indent=1)
print(json.dumps({k: {c: v.get(c) for c in ("SQ_INSTS_VALU", "TCP_TOTAL_CACHE_ACCESSES_sum", "FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "n")} for k, v in res.items()}, indent=1))
PY
# the raw per-pass outputs are tens of MB (gpurun brings back at most 64 MiB of gpurun_out/): the summaries stay
rm -rf $OUT/pmc_plain_[0-9] $OUT/pmc_weighted_[0-9] $OUT/trace_plain $OUT/trace_weighted $OUT/pmc_peaks $OUT/lcp_score.s
