REPO=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/cs_timing_probe.sh > /dev/null 2>&1
mapfile -t ARGS < $REPO/gpurun_out/cs_timing/args.txt
for e in "SHIM_TEST_FRAME_THREADS=1" "SHIM_TEST_FRAME_THREADS=files" "X=1" "PGP_SHIM_FRAME_SERIAL=1"; do
  echo "$e: $(env $e SHIM_TEST_FRAME=3 PGP_SHIM_PRIVATE_RAND=1 PGP_SHIM_SEED=12345 SHIM_TEST_REPEAT=30 $REPO/shim/test_shim "${ARGS[@]}" 2>/dev/null | python3 -c "
import sys, numpy as np
t=sys.stdin.read().splitlines()
ms=np.array([float(x) for l in t if l.startswith('FRAME_MS') for x in l.split()[1:]])[2:]
print([l for l in t if l.startswith('FRAME_SAME')][0], 'median %.3f ms per frame' % np.median(ms))")"
done
