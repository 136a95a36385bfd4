#!/bin/bash
# Build another commit of this repository (library + its own Python package + step_time.py) into
# tools/ab/tree_old/, so that a gpurun call can time it next to the working tree on ONE box:
#   bash tools/ab_commit.sh <rev>            (build container)
#   gpurun -- 'python tools/step_time.py; (cd tools/ab/tree_old && python tools/step_time.py)'
# Box-to-box differences are ~3 %: never compare step times taken in different gpurun calls.
set -e
rev=${1:?usage: ab_commit.sh <rev>}
root=$(cd "$(dirname "$0")/.." && pwd)
wt=$(mktemp -d /tmp/ab_wt.XXXXXX)
git -C "$root" worktree add -f "$wt" "$rev" > /dev/null
make -C "$wt/physimglobalpose_amd/csrc" -j4 > /dev/null
rm -rf "$root/tools/ab/tree_old"
mkdir -p "$root/tools/ab/tree_old/physimglobalpose_amd" "$root/tools/ab/tree_old/tools"
cp "$wt"/physimglobalpose_amd/*.py "$wt/physimglobalpose_amd/libpgp.so" "$root/tools/ab/tree_old/physimglobalpose_amd/"
cp "$wt/tools/step_time.py" "$root/tools/ab/tree_old/tools/"
git -C "$root" worktree remove --force "$wt"
echo "tools/ab/tree_old = $rev"
