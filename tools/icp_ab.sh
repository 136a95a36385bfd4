#!/bin/bash
# A/B of several builds of the library on ONE box: tools/icp_ab.sh a.so b.so ...  (alternating, 3 rounds; summary at the end)
log=${ICP_AB_LOG:-gpurun_out/icp_ab.log}
: > $log
for round in 1 2 3; do
  for lib in "$@"; do
    echo "== $(basename $lib) round $round" >> $log
    PGP_LIB=$lib timeout -k 10 120 python tools/icp_quick.py 10 5 2>&1 | grep -E "poses +(64|256|1024)" >> $log || exit 1
    PGP_LIB=$lib timeout -k 10 120 python tools/icp_config2.py 2>&1 | grep -E "20 reps|resident" >> $log || exit 1
  done
done
python - <<PY
import re,collections
d=collections.defaultdict(list)
lib=None
for l in open("$log"):
    m=re.match(r"== (\S+) round",l)
    if m: lib=m.group(1).replace("libpgp_","").replace(".so",""); continue
    m=re.match(r"(.*?poses +\d+): +([\d.]+) ms/call",l)
    if m: d[(m.group(1).strip(),lib)].append(float(m.group(2))); continue
    m=re.match(r"(host-pointer call|device call, resident index \(token\)): ([\d.]+) ms",l)
    if m: d[("config2 "+m.group(1)[:12],lib)].append(float(m.group(2)))
keys=sorted(set(k for k,_ in d))
libs=[]
for _,l in d:
    if l not in libs: libs.append(l)
print(" "*40+"".join(f"{l:>14s}" for l in libs))
for k in keys:
    print(f"{k:40s}"+"".join(f"{min(d[(k,l)]):14.3f}" for l in libs))
PY
