#!/bin/bash
# VGPR / SGPR / scratch / LDS / occupancy of every kernel in the library (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: scripts/kernel_resources.sh [extra hipcc flags, e.g. -DMMHN_TSB=512]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c -Wno-comment -Rpass-analysis=kernel-resource-usage "$@" \
  -o /tmp/kres.o metmhn_amd/csrc/engine.hip 2>&1 | python3 -c "
import sys,re
cur=None;rows={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'remark: .*?\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)',l)
    if m and cur: rows[cur][m.group(1).split()[0]]=int(m.group(2))
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()
    name=re.sub(r'\(.*','',name)[:70]
    print(f\"{name:72s} vgpr {v.get('VGPRs',0):4d} sgpr {v.get('SGPRs',0):4d} scratch {v.get('ScratchSize',0):4d} occ {v.get('Occupancy',0)} lds {v.get('LDS',0)}\")
"
