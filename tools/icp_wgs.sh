#!/bin/bash
# workgroups per pose (PGP_ICP_WGS) and the solo threshold at 64 poses, all regimes + the config2 call
for cfg in "PGP_ICP_WGS=1" "PGP_ICP_WGS=2" "PGP_ICP_WGS=4" "PGP_ICP_WGS=4 PGP_ICP_SOLO_TICKS=2500" "PGP_ICP_WGS=4 PGP_ICP_SOLO_TICKS=5000"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python tools/icp_quick.py 10 5 2>&1 | grep -E "poses +(8|64) *:" || exit 1
  env $cfg timeout -k 10 120 python tools/icp_config2.py 2>&1 | grep -E "20 reps|resident" || exit 1
done
