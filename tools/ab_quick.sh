#!/bin/bash
# parity subset + A/B of the scoring step against tools/ab/libpgp_base.so (the library built from another
# commit: `git worktree add /tmp/base <rev> && make -C /tmp/base/physimglobalpose_amd/csrc`).  GPU box only.
out=${1:-gpurun_out/ab}
mkdir -p $out
set -o pipefail
python -m pytest tests/test_lcp_gpu.py tests/test_golden_gpu.py tests/test_edge_gpu.py tests/test_stress_gpu.py -x -q 2>&1 | tail -3 | tee $out/parity8.log
grep -q "passed" $out/parity8.log || exit 1
grep -q "failed\|error" $out/parity8.log && exit 1
for round in 1 2; do
  python tools/step_time.py 2>/dev/null | grep -v graph | sed "s/^/new r$round /" | tee -a $out/ab.log
done
cp physimglobalpose_amd/libpgp.so /tmp/libpgp_new.so
cp tools/ab/libpgp_base.so physimglobalpose_amd/libpgp.so
python tools/step_time.py 2>/dev/null | grep -v graph | sed "s/^/base /" | tee -a $out/ab.log
cp /tmp/libpgp_new.so physimglobalpose_amd/libpgp.so
