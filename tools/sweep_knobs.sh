#!/bin/bash
# step time against the launch / index knobs of the scoring kernel (GPU box only)
out=${1:-gpurun_out/sweep}
mkdir -p $out
for hpb in 6 8 10 12; do
  PGP_HPB=$hpb python tools/step_time.py 2>/dev/null | grep -v graph | sed "s/^/hpb=$hpb /" | tee -a $out/sweep.log
done
for tp in 0 5 20; do
  PGP_TAIL_PCT=$tp python tools/step_time.py 2>/dev/null | grep -v graph | sed "s/^/tail_pct=$tp /" | tee -a $out/sweep.log
done
for cr in 0.6 0.7 1.0; do
  PGP_CELL_RATIO=$cr python tools/step_time.py 2>/dev/null | grep -v graph | sed "s/^/cell_ratio=$cr /" | tee -a $out/sweep.log
done
