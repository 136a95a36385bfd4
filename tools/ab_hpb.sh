#!/bin/bash
# step time of one library against the hypotheses-per-workgroup knob (PGP_HPB) -- how much of a step is
# per-workgroup fixed cost.  Usage: bash tools/ab_hpb.sh <lib.so> <out_dir>
lib=$1; out=${2:-gpurun_out/ab}
mkdir -p $out
cp physimglobalpose_amd/libpgp.so /tmp/libpgp_keep.so
cp $lib physimglobalpose_amd/libpgp.so
for hpb in 4 8 16 32 64; do
  PGP_HPB=$hpb PGP_TAIL_PCT=${TAIL:-10} python tools/step_time.py 2>/dev/null | grep -v graph | sed "s/^/$(basename $lib .so) hpb=$hpb /" | tee -a $out/hpb.log
done
cp /tmp/libpgp_keep.so physimglobalpose_amd/libpgp.so
