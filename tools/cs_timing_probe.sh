#!/bin/bash
# host time of the stages of pgp_find_congruent_batch inside a drop-in call (PGP_CS_TIMING=1: each stage then ends with a
# stream synchronisation, so the device stages read longer than in a normal call; "rows + cones (host)" is host work only)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/cs_timing
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
PGP_CS_TIMING=1 SHIM_TEST_INMEMORY=1 PGP_SHIM_SEED=12345 SHIM_TEST_REPEAT=8 $REPO/shim/test_shim "${ARGS[@]}" 2>&1 >/dev/null | grep "congruent batch" | tail -7
