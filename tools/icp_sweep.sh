#!/bin/bash
# A/B sweep of the ICP knobs on the default path (tools/icp_quick.py): vicinity graph off / hops 1..4, solo threshold
out=${1:-gpurun_out/icp_sweep.log}
: > $out
for cfg in "PGP_ICP_VIC=0" "PGP_ICP_HOPS=1" "PGP_ICP_HOPS=2" "PGP_ICP_HOPS=3" "PGP_ICP_HOPS=4" "PGP_ICP_HOPS=2 PGP_ICP_SOLO_TICKS=600" "PGP_ICP_HOPS=2 PGP_ICP_SOLO_TICKS=2000" "PGP_ICP_HOPS=2 PGP_ICP_WGS=2"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 120 python tools/icp_quick.py 10 5 2>&1 | grep -v amdgpu.ids | grep -E "poses +(64|256|1024)" >> $out || exit 1
done
