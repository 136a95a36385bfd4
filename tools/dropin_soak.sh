#!/bin/bash
# soak of the drop-in at a fixed seed: every call of a long run must return exactly what the first one did (in-memory overload
# and file hand-off, one object and two objects alternating) -- a race between a call's asynchronous parts (the index build on
# its side stream, the PNG decoder thread) would show as a call that differs
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/dropin_soak
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
N=${1:-2000}
for MODE in "SHIM_TEST_INMEMORY=1" "SHIM_TEST_INMEMORY=1 SHIM_TEST_TWO_OBJECTS=1" "SHIM_TEST_FILES=1"; do
  echo "== $MODE, $N calls"
  env $MODE SHIM_TEST_CHECK_SAME=1 PGP_SHIM_SEED=12345 SHIM_TEST_REPEAT=$N $REPO/shim/test_shim "${ARGS[@]}" 2>/dev/null | grep "SAME_AS_FIRST\|BEST_SCORE"
done
# frames of three objects, side by side: through the frame entry point, through a fresh thread per object around the single call
# (in memory and through the files), and every job of a frame the SAME object (the calls take turns on its context)
for MODE in "SHIM_TEST_FRAME=3" "SHIM_TEST_FRAME=3 SHIM_TEST_FRAME_THREADS=1" "SHIM_TEST_FRAME=3 SHIM_TEST_FRAME_THREADS=files" "SHIM_TEST_FRAME=4 SHIM_TEST_FRAME_ONE_OBJECT=1"; do
  M=$((N / 4))
  echo "== $MODE, $M frames"
  env $MODE PGP_SHIM_PRIVATE_RAND=1 PGP_SHIM_SEED=12345 SHIM_TEST_REPEAT=$M $REPO/shim/test_shim "${ARGS[@]}" 2>/dev/null | grep "FRAME_SAME\|BEST_SCORE"
done
