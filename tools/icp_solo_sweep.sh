#!/bin/bash
# sweep of the share -> solo threshold of the clustered ICP launch (100 MHz ticks), plus the one-workgroup launch
PGP_ICP_WGS=1 python tools/icp_regimes.py "wgs 1" 2>/dev/null
for st in 0 600 1100 1600 2500 4000 1000000000; do
  PGP_ICP_SOLO_TICKS=$st python tools/icp_regimes.py "solo_ticks $st" 2>/dev/null
done
