#!/bin/bash
# Print VGPR / SGPR-spill / scratch / occupancy of every kernel in csrc/dcc_env.hip (compile-only, no GPU).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Iinclude -c \
  dynamic-coverage-control_amd/csrc/dcc_env.hip -o /tmp/_res.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    if 'error' in l: print(l.strip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':re.sub(r'.*dcc_env_kernelI(.*)EEvNS.*',r'\1',m.group(1))}; rows.append(cur); continue
    for k,pat in (('vgpr',r' VGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('sspill',r'SGPRs Spill: (\d+)'),('vspill',r'VGPRs Spill: (\d+)')):
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=int(m.group(1))
for r in rows:
    flag='  <-- SCRATCH' if r.get('scratch',0) else ''
    print('%-28s vgpr=%3d occ=%d sgpr_spill=%3d vgpr_spill=%3d scratch=%4d%s'%(r['name'],r.get('vgpr',0),r.get('occ',0),r.get('sspill',0),r.get('vspill',0),r.get('scratch',0),flag))
"
