#!/bin/bash
# VGPRs / scratch / LDS / occupancy of the kernels of one translation unit:  bash tools/kernel_regs.sh feature.hip [name filter]
cd "$(dirname "$0")/../multi-modal-loam_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I../../include -I. $EXTRA \
  -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/kernel_regs_$$.o 2>&1 | python3 -c "
import re,sys,subprocess
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur=subprocess.run(['/usr/bin/c++filt',m.group(1)],capture_output=True,text=True).stdout.strip().replace('(anonymous namespace)::','').split('(')[0]; rows[cur]={}
    for k in ('VGPRs','AGPRs','ScratchSize [bytes/lane]','Occupancy [waves/SIMD]','LDS Size [bytes/block]','SGPRs'):
        m=re.search(re.escape(k)+r': (\d+)',l)
        if m and cur: rows[cur][k]=int(m.group(1))
flt=sys.argv[1] if len(sys.argv)>1 else ''
for k,v in rows.items():
    if flt in k: print('%-60s vgpr %3d sgpr %3d scratch %4d lds %6d occ %d'%(k[:60],v.get('VGPRs',0),v.get('SGPRs',0),v.get('ScratchSize [bytes/lane]',0),v.get('LDS Size [bytes/block]',0),v.get('Occupancy [waves/SIMD]',0)))
" "$2"
rm -f /tmp/kernel_regs_$$.o
