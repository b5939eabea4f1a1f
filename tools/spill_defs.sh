#!/bin/bash
# For one physics_ll_kernel instantiation: every scratch STORE with the source line of the instruction that defined the stored register, and the
# scratch LOADS by frame offset (which spilled value is reloaded how often, statically).  usage: tools/spill_defs.sh <8 template flags> [extra flags]
T=$1; shift
cd "$(dirname "$0")/../vid2player3d_amd/csrc"
M=$(echo $T | sed 's/./Lb&E/g')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast-honor-pragmas -fno-vectorize -fno-slp-vectorize -fassociative-math -freciprocal-math -fno-signed-zeros -fno-trapping-math -fno-honor-nans -mllvm -sink-insts-to-avoid-spills=1 "$@" -gline-tables-only --cuda-device-only -S physics_ll.hip -o /tmp/spilld_$$.s 2>/dev/null
python3 - /tmp/spilld_$$.s "_ZN3v2p17physics_ll_kernelI${M}EEvNS_8PhysArgsE" <<'PY'
import re,sys,collections
L=open(sys.argv[1]).read().split('\n'); name=sys.argv[2]
s=next(i for i,l in enumerate(L) if l.startswith(name+':')); e=next(i for i in range(s,len(L)) if L[i].startswith('.Lfunc_end'))
cur=None; lastdef={}; n=0; stores=[]; loads=collections.Counter(); pos=collections.defaultdict(list)
for i in range(s,e):
    l=L[i]
    m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)\s+(\d+)',l)
    if m: cur=(int(m.group(1)),int(m.group(2))); continue
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    n+=1
    mm=re.match(r'(\S+)\s+(.*)',t)
    if not mm: continue
    op,args=mm.group(1),mm.group(2)
    off=re.search(r'offset:(\d+)',args); off=int(off.group(1)) if off else 0
    if op.startswith('scratch_store'):
        data=[p.strip() for p in args.split(',')][1]
        r=re.match(r'v\[?(\d+)',data)
        stores.append((n,off,op.replace('scratch_store_',''),lastdef.get(int(r.group(1))) if r else None))
    elif op.startswith('scratch_load'):
        loads[off]+=1; pos[off].append(n)
    else:
        first=args.split(',')[0].strip()
        r=re.match(r'v\[(\d+):(\d+)\]',first)
        if r:
            for k in range(int(r.group(1)),int(r.group(2))+1): lastdef[k]=(cur,op)
        else:
            r=re.match(r'v(\d+)$',first)
            if r: lastdef[int(r.group(1))]=(cur,op)
print("instructions",n,"stores",len(stores),"loads",sum(loads.values()))
for st in stores: print("store @%5d offset %3d %-8s defined by %s | reloaded %d x at %s"%(st[0],st[1],st[2],st[3],loads.get(st[1],0),pos.get(st[1],[])[:12]))
PY
rm -f /tmp/spilld_$$.s
