#!/bin/bash
# probe: what makes a call through getProbableTransformsSuper4PCSFrame (one job, caller's thread) slower than the direct call?
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
run() { echo "$1: $(env $2 PGP_SHIM_SEED=12345 SHIM_TEST_REPEAT=16 $REPO/shim/test_shim "${ARGS[@]}" 2>/dev/null | grep "FRAME_MS\|ELAPSED" | cut -c1-130)"; }
run "in-memory                 " "SHIM_TEST_INMEMORY=1 PGP_SHIM_PRIVATE_RAND=1"
run "frame of 1                " "SHIM_TEST_FRAME=1 PGP_SHIM_PRIVATE_RAND=1"
run "frame of 1, same table    " "SHIM_TEST_FRAME=1 PGP_SHIM_PRIVATE_RAND=1 SHIM_TEST_FRAME_SAME_TABLE=1"
run "frame loop, direct call   " "SHIM_TEST_FRAME=1 PGP_SHIM_PRIVATE_RAND=1 SHIM_TEST_FRAME_DIRECT=1"
run "direct call, same table   " "SHIM_TEST_FRAME=1 PGP_SHIM_PRIVATE_RAND=1 SHIM_TEST_FRAME_DIRECT=1 SHIM_TEST_FRAME_SAME_TABLE=1"
