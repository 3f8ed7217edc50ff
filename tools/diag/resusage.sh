#!/bin/bash
# usage: resusage.sh file.hip [extra flags] -> kernel name, VGPRs, AGPRs, scratch, occupancy, LDS
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f -o /tmp/ru_$$.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for key,pat in (('vgpr',r' VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)'),('sgpr',r' SGPRs: (\d+)')):
        m=re.search(pat,l)
        if m and cur is not None: cur[key]=int(m.group(1))
import subprocess
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=re.sub(r'\(.*','',n)
    print(f\"{n[:70]:70s} v{r.get('vgpr',0):4d} a{r.get('agpr',0):4d} scratch{r.get('scratch',0):5d} occ{r.get('occ',0):2d} lds{r.get('lds',0):7d}\")
"
rm -f /tmp/ru_$$.o
