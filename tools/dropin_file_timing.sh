#!/bin/bash
# the file hand-off of the drop-in (three ASCII PLYs + the probability PNG read on four threads): what each part takes
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/file_timing
mkdir -p $OUT
python3 - "$OUT" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from _dropin import make_dropin_case
d = os.path.join(sys.argv[1], "case"); os.makedirs(d, exist_ok=True)
args, case = make_dropin_case(d, png=os.environ.get("DROPIN_PNG", "pil"))   # pil | opencv | opencv_noisy (tests/_dropin.py)
open(os.path.join(sys.argv[1], "args.txt"), "w").write("\n".join(args))
PY
mapfile -t ARGS < $OUT/args.txt
PGP_SHIM_VERBOSE=1 PGP_SHIM_SEED=12345 SHIM_TEST_REPEAT=10 $REPO/shim/test_shim "${ARGS[@]}" > $OUT/out.txt 2> $OUT/err.txt
grep "file hand-off" $OUT/err.txt | tail -4
grep "PHASES" $OUT/err.txt | tail -2
grep ELAPSED $OUT/out.txt | tail -3
ls -la $OUT/case
