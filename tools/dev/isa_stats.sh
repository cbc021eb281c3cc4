#!/bin/bash
# usage: tools/dev/isa_stats.sh <file.hip> [kernel-name-regex]   — compiles one source for gfx950 and prints per-kernel
# register counts, spill instructions (v_readlane / v_writelane of spilled scalars, scratch) and instruction totals
set -e
SRC=$1; PAT=${2:-.}
CS=/root/repo/tokengeex_amd/csrc
mkdir -p /tmp/isa && cd /tmp/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-result -Wno-pass-failed -x hip -S $CS/$SRC -o /tmp/isa/$SRC.s --cuda-device-only 2>&1 | grep -v "^$" | grep -v "hip-link" | head -20
python3 - "$SRC" "$PAT" <<'PY'
import re,collections,sys
s=open('/tmp/isa/'+sys.argv[1]+'.s').read()
pat=re.compile(sys.argv[2])
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name=m.group(1)
    if not pat.search(name): continue
    i=s.index(name+':'); j=s.index('s_endpgm',i)
    cnt=collections.Counter()
    for l in s[i:j].split('\n'):
        l=l.strip()
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'): continue
        cnt[l.split()[0]]+=1
    b=m.group(2)
    g=lambda k: re.search(k+r' (\d+)',b).group(1)
    valu=sum(v for k,v in cnt.items() if k.startswith('v_'))
    print(name[:70],'| total',sum(cnt.values()),'valu',valu,'salu',sum(v for k,v in cnt.items() if k.startswith('s_')),'readlane',cnt['v_readlane_b32'],'writelane',cnt['v_writelane_b32'],'vgpr',g('next_free_vgpr'),'sgpr',g('next_free_sgpr'),'scratch',g('private_segment_fixed_size'))
PY
