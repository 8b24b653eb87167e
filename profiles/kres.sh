#!/bin/bash
# kernel resource usage of hrt_kernels.hip (VGPRs, SGPRs, scratch, occupancy) -- compile-only, no GPU
cd "$(dirname "$0")/../hermespy-rt_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -fno-slp-vectorize -I../include -Icsrc $EXTRA \
  -c csrc/hrt_kernels.hip -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass', l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name') or t.startswith('Name:'):
        if cur: print(cur)
        n=t.split(':',1)[1].strip()
        n=re.sub(r'_ZN12_GLOBAL__N_1\d+','',n)
        cur=n[:40].ljust(42)
    elif any(t.startswith(k) for k in ('VGPRs:','SGPRs:','TotalSGPRs','ScratchSize','Occupancy','LDS Size')):
        cur+=' '+t.replace(' [bytes/lane]','').replace(' [waves/SIMD]','').replace(' [bytes/block]','')
if cur: print(cur)
"
