// Does a gfx950 SIMD skip the passes of a wave64 VALU instruction whose lanes are masked off?  Dependent and independent v_fma_f32 streams
// with EXEC = all 64 lanes / the lower 32 / the lower 16, one wave per SIMD.  (If half-masked instructions took half the time, a heavy
// env alone in its wave on lanes 0..31 would run its serial chain twice as fast.)
// Build: hipcc --offload-arch=gfx950 -O3 -o exec_mask exec_mask.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int ITER = 4096;

template <int DEP> __global__ void k(float* out, float seed, int active) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    if ((int)threadIdx.x < active) {
        for (int i = 0; i < ITER; ++i) {
            if (DEP) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a0 = a0 * m + c;
            } else {
                a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c; a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int DEP> void run(const char* name, int active, int wps, float* d) {
    const int blocks = 256 * 4 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<DEP>, dim3(blocks), dim3(64), 0, 0, d, 1.f, active);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<DEP>, dim3(blocks), dim3(64), 0, 0, d, 1.f, active);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s EXEC = lower %2d lanes, %d wave(s)/SIMD: %.3f ms -> %.2f cycles per wave-instruction at 2.4 GHz\n", name, active, wps, ms, ms * 1e-3 * 2.4e9 / (8.0 * ITER));
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 4 * 64 * sizeof(float));
    for (int wps : {1, 3})
        for (int active : {64, 32, 16}) {
            run<1>("dependent v_fma_f32 chain", active, wps, d);
            run<0>("8 independent v_fma_f32", active, wps, d);
        }
    return 0;
}
