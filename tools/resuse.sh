#!/bin/bash
# Register / LDS / occupancy table of every kernel of one csrc/*.hip file (hipcc's kernel-resource-usage remarks):
#   tools/resuse.sh hit_adv_amd/csrc/pointnet.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -I"$(dirname "$0")/../include" -c "$1" -o /tmp/_resuse.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r' Name: (\S+)',l)
    if m: cur={'name':m.group(1)};rows.append(cur);continue
    for k,pat in (('v',r'VGPRs: (\d+)'),('a',r'AGPRs: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)'),('scr',r'ScratchSize \[bytes/lane\]: (\d+)')):
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=m.group(1)
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.splitlines()
for r,n in zip(rows,names):
    print('%-72s v=%s a=%s occ=%s lds=%s scratch=%s'%(re.sub(r'\(.*','',n)[:72],r.get('v'),r.get('a'),r.get('occ'),r.get('lds'),r.get('scr')))
"
