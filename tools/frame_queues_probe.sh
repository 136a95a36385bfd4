#!/bin/bash
# probe: the runtime's limit on hardware queues per process (GPU_MAX_HW_QUEUES, default 4) against contexts that each own
# a stream and a side stream for their index build: one call on a second context, and frames of several objects
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/frame
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
for Q in default 8 16; do
  for OBJ in 1 3 6; do
    for MODE in side_by_side one_by_one; do
      EXTRA=""; [ $MODE = one_by_one ] && EXTRA="PGP_SHIM_FRAME_SERIAL=1"
      [ $Q != default ] && EXTRA="$EXTRA GPU_MAX_HW_QUEUES=$Q"
      env $EXTRA SHIM_TEST_FRAME=$OBJ PGP_SHIM_PRIVATE_RAND=1 PGP_SHIM_SEED=12345 SHIM_TEST_REPEAT=30 $REPO/shim/test_shim "${ARGS[@]}" 2>/dev/null > $OUT/out.txt || { echo "failed"; exit 1; }
      python3 - "$OUT/out.txt" "$OBJ" "$MODE" "$Q" <<'PY'
import sys, numpy as np
t = open(sys.argv[1]).read().splitlines()
ms = np.array([float(x) for l in t if l.startswith("FRAME_MS") for x in l.split()[1:]])[2:]
same = [l for l in t if l.startswith("FRAME_SAME")][0]
print(f"queues {sys.argv[4]:7s} {sys.argv[2]} objects, {sys.argv[3]:12s}: median {np.median(ms):.3f} ms per frame (min {ms.min():.3f}, p90 {np.percentile(ms, 90):.3f})  {same}")
PY
    done
  done
done
