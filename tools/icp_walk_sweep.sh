#!/bin/bash
# first-iteration graph walk (PGP_ICP_FIRST_WALK = moves): 0 (off) .. 5, all regimes + the config2 call
for w in 0 1 2 3 5; do
  echo "== PGP_ICP_FIRST_WALK=$w"
  PGP_ICP_FIRST_WALK=$w timeout -k 10 120 python tools/icp_quick.py 10 5 2>&1 | grep -E "poses +(64|256|1024)" || exit 1
  PGP_ICP_FIRST_WALK=$w timeout -k 10 120 python tools/icp_config2.py 2>&1 | grep -E "20 reps|resident" || exit 1
done
