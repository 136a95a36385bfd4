#!/bin/bash
# A/B of the one-launch scene-sized ICP (csrc/icp.hip icp_scene_persist) on ONE box: workgroups, poll interval, and the
# fixed part of a call (1 iteration) for both paths.   usage: bash tools/icp_scene_sweep.sh [log]
log=${1:-gpurun_out/icp_scene_sweep.log}
mkdir -p $(dirname $log)
run() { echo -n "$* : " | tee -a $log; env "$@" python tools/icp_table.py 2>/dev/null | tail -1 | tee -a $log; }
run ICP_TABLE_ITERS=1 PGP_ICP_SCENE_PERSIST=0
run ICP_TABLE_ITERS=1 X=1
run PGP_ICP_SCENE_PERSIST=0
run X=1
for w in 257 513 1025; do for s in 0 1 4; do run PGP_ICP_SCENE_WGS=$w PGP_ICP_SCENE_SLEEP=$s; done; done
