#!/bin/bash
# compact table: template flags (CONTACT MULTI TGS DIAG BALL JOBS LIMITS) / VGPRs / scratch bytes per lane / occupancy of every physics_ll_kernel
# instantiation as the compiler reports it (CPU only).  Usage: tools/kres.sh [extra -D flags]
cd "$(dirname "$0")/../vid2player3d_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast-honor-pragmas -fno-vectorize -fno-slp-vectorize -fassociative-math -freciprocal-math -fno-signed-zeros -fno-trapping-math -fno-honor-nans "$@" -Rpass-analysis=kernel-resource-usage -c physics_ll.hip -o /tmp/kres_$$.o 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: [^:]*:[0-9]*:[0-9]*: //; s/\[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - |
  grep physics_ll_kernel | sed 's/Function Name: _ZN3v2p17physics_ll_kernelI//; s/EEvNS_8PhysArgsE//; s/Lb//g; s/E/ /g' | sed "s/physics_ll.hip:[0-9]*:[0-9]*: remark: //g; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/" | tr -s " \t" " "
rm -f /tmp/kres_$$.o
