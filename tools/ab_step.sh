#!/bin/bash
# A/B of the scoring step on ONE box: every tools/ab/libpgp_*.so (built from other commits or with
# -DPGP_ABLATE=n timing experiments) against the in-tree library.  Usage (on the GPU box):
#   bash tools/ab_step.sh [out_dir] [rounds]
out=${1:-gpurun_out/ab}
rounds=${2:-2}
mkdir -p $out
cp physimglobalpose_amd/libpgp.so /tmp/libpgp_new.so
for round in $(seq 1 $rounds); do
  for lib in /tmp/libpgp_new.so tools/ab/libpgp_*.so; do
    name=$(basename $lib .so | sed 's/libpgp_//')
    cp $lib physimglobalpose_amd/libpgp.so
    python tools/step_time.py 2>/dev/null | grep -v graph | sed "s/^/$name r$round /" | tee -a $out/ab.log
  done
done
cp /tmp/libpgp_new.so physimglobalpose_amd/libpgp.so
