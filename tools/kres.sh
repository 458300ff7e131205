#!/bin/bash
# register / spill / occupancy table of every lgd_scan_kernel variant (compile only, no GPU)
cd "$(dirname "$0")/../loudgain_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c lgd_kernels.hip -o /tmp/lgd_k.o \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re
txt=sys.stdin.read()
blocks=re.split(r'remark: [^\n]*Function Name: ',txt)[1:]
for b in blocks:
    name=b.split('\n')[0]
    m=re.search(r'lgd_scan_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)',name)
    def g(k):
        r=re.search(k+r': (\d+)',b); return r.group(1) if r else '?'
    if m: print('C=%3s G=%s TP=%s W=%s'%m.groups(), 'VGPR',g('VGPRs'),'SGPR',g('SGPRs'),'spill',g('VGPRs Spill'),'scratch',g(r'ScratchSize \[bytes/lane\]'),'occ',g(r'Occupancy \[waves/SIMD\]'))
" | sort
