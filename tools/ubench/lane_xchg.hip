// Latency of cross-lane moves on gfx950, dependent chains (one wave per SIMD and 2 waves per SIMD): DPP row_shr / wave_shr / wave_shl,
// ds_bpermute, v_readlane+v_writelane.  Build: hipcc --offload-arch=gfx950 -O3 -o lane_xchg lane_xchg.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int ITER = 2048;
template <int MODE> __global__ void k(float* out, int idx) {
    float a = threadIdx.x * 0.5f + 1.f;
    const int src = ((threadIdx.x + 63) & 63) << 2;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int v = __float_as_int(a);
            if (MODE == 0) v = __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false);        // row_shr:1
            else if (MODE == 1) v = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
            else if (MODE == 2) v = __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
            else if (MODE == 3) v = __builtin_amdgcn_ds_bpermute(src, v);
            else if (MODE == 4) v = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1,3
            else if (MODE == 5) v = __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);           // quad_perm [1,0,3,2]
            a = __int_as_float(v) * 1.0000001f;  // one dependent VALU op between the moves
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + idx;
}
template <int MODE> void run(const char* name, int wps, float* d) {
    const int blocks = 256 * 4 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 0); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 1); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s waves/SIMD %d: %.3f ms -> %.1f cycles (2.4 GHz nominal) per {move + 1 dependent v_mul}\n", name, wps, ms, ms * 1e-3 * 2.4e9 / (8.0 * ITER));
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 4 * 4 * 64 * sizeof(float));
    for (int w : {1, 2}) {
        run<0>("dpp row_shr:1", w, d); run<5>("dpp quad_perm", w, d); run<4>("dpp row_bcast:15", w, d); run<1>("dpp wave_shr:1", w, d);
        run<2>("dpp wave_shl:1", w, d); run<3>("ds_bpermute_b32", w, d);
    }
    return 0;
}
