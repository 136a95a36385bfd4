#!/bin/bash
# the outlier ICP case on this round's library, HEAD's icp.hip, and the round-4 build
set -e
mkdir -p gpurun_out
echo "== tree"; timeout -k 10 200 python tools/icp_outlier_probe.py 8
echo "== head icp.hip"; PGP_LIB=$PWD/tools/ab/libpgp_head.so timeout -k 10 200 python tools/icp_outlier_probe.py 8
echo "== round 4"; PGP_PKG_ROOT=$PWD/tools/ab/r4 timeout -k 10 200 python tools/icp_outlier_probe.py 8
echo "== tree, debug"; PGP_ICP_DEBUG=1 timeout -k 10 200 python tools/icp_outlier_probe.py 2 256 2>&1 | tail -20
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5_outl -o outl -- python3 $GRAFT_REPO_ROOT/tools/icp_outlier_probe.py 4 256 > $GRAFT_REPO_ROOT/gpurun_out/r5_outl.log 2>&1 || true
find $GRAFT_REPO_ROOT/gpurun_out/r5_outl -name '*kernel_stats.csv' -exec head -8 {} \;
