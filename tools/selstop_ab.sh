#!/bin/bash
# timing experiment: select_bases cut short after point 2 / 3 / 4 (tools/ab/libpgp_selstop<k>.so, wrong results) under the
# drop-in's kernel trace -- where the kernel's ~110 us go
cp physimglobalpose_amd/libpgp.so /tmp/libpgp_keep.so
for k in 2 3 4; do
  cp tools/ab/libpgp_selstop$k.so physimglobalpose_amd/libpgp.so
  bash tools/dropin_trace.sh > /tmp/trace_$k.txt 2>&1
  echo "stop after point $k: $(grep select_bases /tmp/trace_$k.txt)"
done
cp /tmp/libpgp_keep.so physimglobalpose_amd/libpgp.so
bash tools/dropin_trace.sh > /tmp/trace_full.txt 2>&1
echo "whole kernel:       $(grep select_bases /tmp/trace_full.txt)"
