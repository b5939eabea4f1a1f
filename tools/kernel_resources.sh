#!/bin/bash
# register / scratch / LDS usage of every kernel of one translation unit (default physics_ll.hip), as the compiler reports it
F=${1:-physics_ll.hip}; cd "$(dirname "$0")/../vid2player3d_amd/csrc"
EXTRA=""; [ "$F" = physics_ll.hip ] && EXTRA="-fassociative-math -freciprocal-math -fno-signed-zeros -fno-trapping-math -fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast-honor-pragmas -fno-vectorize -fno-slp-vectorize $EXTRA ${V2P_EXTRA_FLAGS:-} -Rpass-analysis=kernel-resource-usage -c $F -o /tmp/kr_$$.o 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed 's/.*remark: [^:]*:[0-9]*:[0-9]*: //' | paste - - - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | tr -s ' '
rm -f /tmp/kr_$$.o
