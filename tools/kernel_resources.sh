#!/bin/bash
# usage: tools/kernel_resources.sh physimglobalpose_amd/csrc/icp.hip [extra hipcc flags]
# prints one line per kernel: VGPRs, SGPRs, spills, LDS, occupancy (from -Rpass-analysis=kernel-resource-usage)
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -c "$f" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)",l)
    if not m: continue
    k,v=m.groups()
    if k=="Function Name":
        if cur: print(cur)
        cur=v[:70].ljust(72)
    else: cur+=f" {k.split()[0]}={v}"
if cur: print(cur)'
