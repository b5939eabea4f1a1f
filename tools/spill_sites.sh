#!/bin/bash
# where a physics_ll_kernel instantiation spills: scratch stores / loads by source line.  usage: tools/spill_sites.sh <8 template flags CONTACT MULTI TGS DIAG BALL JOBS LIMITS VFRIC, e.g. 10001110> [extra flags]
T=$1; shift
cd "$(dirname "$0")/../vid2player3d_amd/csrc"
M=$(echo $T | sed 's/./Lb&E/g')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast-honor-pragmas -fno-vectorize -fno-slp-vectorize -fassociative-math -freciprocal-math -fno-signed-zeros -fno-trapping-math -fno-honor-nans -mllvm -sink-insts-to-avoid-spills=1 "$@" -gline-tables-only --cuda-device-only -S physics_ll.hip -o /tmp/spill_$$.s 2>/dev/null
python3 - /tmp/spill_$$.s "_ZN3v2p17physics_ll_kernelI${M}EEvNS_8PhysArgsE" <<'PY'
import re,sys,collections
L=open(sys.argv[1]).read().split('\n'); name=sys.argv[2]
s=next(i for i,l in enumerate(L) if l.startswith(name+':')); e=next(i for i in range(s,len(L)) if L[i].startswith('.Lfunc_end'))
cur=None; st=collections.Counter(); ld=collections.Counter(); wl=collections.Counter(); rl=collections.Counter()
for l in L[s:e]:
    m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)',l)
    if m: cur=(int(m.group(1)),int(m.group(2)))
    if 'scratch_store' in l: st[cur]+=1
    if 'scratch_load' in l: ld[cur]+=1
    if 'v_writelane' in l: wl[cur]+=1
    if 'v_readlane' in l: rl[cur]+=1
print("instructions",e-s,"scratch stores",sum(st.values()),"loads",sum(ld.values()),"v_writelane",sum(wl.values()),"v_readlane",sum(rl.values()))
print("stores by (file,line):",sorted(st.items(),key=lambda x:-x[1])[:25])
print("loads by (file,line):",sorted(ld.items(),key=lambda x:-x[1])[:40])
PY
rm -f /tmp/spill_$$.s
