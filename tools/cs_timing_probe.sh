REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/frame
mapfile -t ARGS < $OUT/args.txt
PGP_CS_TIMING=1 SHIM_TEST_INMEMORY=1 PGP_SHIM_SEED=12345 SHIM_TEST_REPEAT=8 $REPO/shim/test_shim "${ARGS[@]}" 2>&1 >/dev/null | grep "congruent batch" | tail -7
