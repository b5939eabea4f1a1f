#!/bin/bash
# compact table: template flags (CONTACT MULTI TGS DIAG BALL JOBS LIMITS) / VGPRs / scratch bytes per lane / occupancy of every physics_ll_kernel
# instantiation as the compiler reports it (CPU only).  Usage: tools/kres.sh [--regs] [extra -D flags]   (--regs: the flags of the library's
# second, register build - vid2player3d_amd/build.py)
if [ "$1" = "--regs" ]; then shift; set -- -Dv2p=v2p_regs -DV2P_LL_WPS=2 -DV2P_LL_WPS_BALL=2 -DV2P_LL_WPS_LIMITS=2 -DV2P_LL_PARK2=0 -DV2P_LL_PARK3=0 -mllvm -amdgpu-sched-strategy=iterative-ilp "$@"; fi
cd "$(dirname "$0")/../vid2player3d_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast-honor-pragmas -fno-vectorize -fno-slp-vectorize -fassociative-math -freciprocal-math -fno-signed-zeros -fno-trapping-math -fno-honor-nans -mllvm -sink-insts-to-avoid-spills=1 "$@" -Rpass-analysis=kernel-resource-usage -c physics_ll.hip -o /tmp/kres_$$.o 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: [^:]*:[0-9]*:[0-9]*: //; s/\[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - |
  grep physics_ll_kernel | sed 's/Function Name: _ZN[0-9]*v2p[_regs]*17physics_ll_kernelI//; s/EEvNS_8PhysArgsE//; s/Lb//g; s/E/ /g' | sed "s/physics_ll.hip:[0-9]*:[0-9]*: remark: //g; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/" | tr -s " \t" " "
rm -f /tmp/kres_$$.o
