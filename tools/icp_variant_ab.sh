#!/bin/bash
# usage: tools/icp_variant_ab.sh NAME  -- ICP equality tests on tools/ab/libpgp_NAME.so, then interleaved icp_quick timings
set -e
V=$PWD/tools/ab/libpgp_$1.so
echo "== ICP tests on $1"
PGP_LIB=$V timeout -k 10 600 python -m pytest tests/test_icp_index_gpu.py tests/test_icp_variants_gpu.py tests/test_icp_gpu.py tests/test_icp_multi_gpu.py -m gpu -q -x 2>&1 | tail -4
for r in 1 2; do
  echo "== default, run $r"; timeout -k 10 200 python tools/icp_quick.py 10 5
  echo "== $1, run $r"; PGP_LIB=$V timeout -k 10 200 python tools/icp_quick.py 10 5
done
