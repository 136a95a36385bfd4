#!/bin/bash
# 512-thread ICP workgroups (256 registers per lane, no spills) against the 1024-thread default: equality of the paths
# inside the 512 build first, then interleaved timings
set -e
T512=$PWD/tools/ab/libpgp_t512.so
echo "== tests on the 512-thread build (its paths against each other; the golden files hold 1024-thread bits)"
PGP_LIB=$T512 timeout -k 10 500 python -m pytest tests/test_icp_index_gpu.py tests/test_icp_variants_gpu.py tests/test_icp_gpu.py -m gpu -q -x 2>&1 | tail -15 || true
for r in 1 2; do
  echo "== default (1024 threads), run $r"; timeout -k 10 200 python tools/icp_quick.py 10 5
  echo "== 512 threads, run $r"; PGP_LIB=$T512 timeout -k 10 200 python tools/icp_quick.py 10 5
done
