#!/bin/bash
# Register / occupancy summary of every kernel of one source:  bash tools/resource_usage.sh node_chain.hip [extra flags]
cd "$(dirname "$0")/../hermnet_amd/csrc"
SRC=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c $SRC -o /dev/null 2>&1 | python3 -c "
import sys,re
cur=None
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m: cur=m.group(1); d={}; continue
    m=re.search(r'remark:\s+([A-Za-z /\[\]]+?): (\d+)',ln)
    if m and cur:
        d[m.group(1).strip()]=m.group(2)
        if m.group(1).strip().startswith('LDS'):
            print('%-64s' % cur[16:80], ' '.join('%s=%s' % (k.split(' [')[0].replace(' ','_'), v) for k, v in d.items() if k.split(' [')[0] in ('VGPRs','AGPRs','VGPRs Spill','SGPRs Spill','Occupancy','ScratchSize')))
"
