// VALU issue model of one gfx950 SIMD: throughput of independent / dependent v_fma_f32 and v_pk_fma_f32 streams at 1, 2, 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 32768;  // (round 6: 8 x longer - ~1 ms kernels, past launch overhead and clock ramp)

// clk (nullable): per wave, [shader-clock cycles (s_memtime), constant 100 MHz ticks (s_memrealtime)] spent in the loop - the clock the
// SIMD actually ran at = cycles / ticks x 100 MHz, and the issue interval in REAL cycles instead of "nominal cycles at an assumed 2.4 GHz"
template <int MODE> __global__ void k(float* out, float seed, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0{a0, a1}, p1{a2, a3}, p2{a4, a5}, p3{a6, a7}, p4{a1, a0}, p5{a3, a2}, p6{a5, a4}, p7{a7, a6};
    const float m = 1.0000001f, c = 1e-9f;
    const f2 m2{m, m}, c2{c, c};
    for (int i = 0; i < ITER; ++i) {
        if (MODE == 0) {  // 8 independent scalar fma chains
            a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c; a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
        } else if (MODE == 1) {  // one dependent chain, 8 per iteration
#pragma unroll
            for (int j = 0; j < 8; ++j) a0 = a0 * m + c;
        } else if (MODE == 2) {  // 8 independent packed chains
            p0 = p0 * m2 + c2; p1 = p1 * m2 + c2; p2 = p2 * m2 + c2; p3 = p3 * m2 + c2; p4 = p4 * m2 + c2; p5 = p5 * m2 + c2; p6 = p6 * m2 + c2; p7 = p7 * m2 + c2;
        } else if (MODE == 3) {  // one dependent packed chain
#pragma unroll
            for (int j = 0; j < 8; ++j) p0 = p0 * m2 + c2;
        } else if (MODE == 4) {  // two dependent chains interleaved
#pragma unroll
            for (int j = 0; j < 4; ++j) { a0 = a0 * m + c; a1 = a1 * m + c; }
        }
    }
    if (clk && threadIdx.x == 0) { clk[2 * blockIdx.x] = clock64() - c0; clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.y;
}

template <int MODE> void run(const char* name, int waves_per_simd, float* d) {
    const int cus = 256, blocks = cus * 4 * waves_per_simd;  // one wave per block: spread over all SIMDs
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    static long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, sizeof(long long) * 2 * 256 * 4 * 8);
    for (int warm = 0; warm < 3; ++warm) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 1.f, nullptr);  // (clock ramp)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 1.f, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_wave = 8.0 * ITER;
    const double cyc = ms * 1e-3 * 2.4e9;  // at 2.4 GHz nominal
    static long long h[2 * 256 * 4 * 8];
    hipMemcpy(h, clk, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double sc = 0, sw = 0;
    for (int i = 0; i < blocks; ++i) { sc += (double)h[2 * i]; sw += (double)h[2 * i + 1]; }
    sc /= blocks; sw /= blocks;
    printf("%-34s waves/SIMD %d: %.3f ms  -> %.2f nominal cycles (2.4 GHz, HIP events) per instruction per SIMD | in-kernel: s_memtime %.0f, s_memrealtime %.0f ticks per wave "
           "-> s_memtime / s_memrealtime = %.3f (x 100 MHz = %.0f MHz if s_memtime counts shader clocks), %.2f s_memtime cycles per instruction per SIMD\n", name, waves_per_simd, ms,
           cyc / (insts_per_wave * waves_per_simd), sc, sw, sc / sw, 100.0 * sc / sw, sc / (insts_per_wave * waves_per_simd));
}

// ---- round 6: controlled residency.  One workgroup per CU (96 KB of dynamic LDS: a second one does not fit), 4 x W waves in it = W waves on
// every SIMD for the whole kernel - the table above launches one-wave workgroups and leaves their placement to the dispatcher (the per-wave
// in-kernel times show them running in 1.3 .. 2 rounds: its "waves/SIMD" is nominal).  Run under
//   rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -- ./valu_issue cu
// the issue interval in the units bench.py quotes for the physics kernel, 4 x SQ_WAVE_CYCLES / W / SQ_INSTS_VALU, needs no clock at all.
template <int MODE> __global__ void kcu(float* out, float seed, long long* clk) {
    extern __shared__ float pad[];
    if (seed == 12345.f) pad[threadIdx.x] = seed;  // (keeps the allocation)
    const long long w0 = wall_clock64();
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < ITER; ++i) {
        if (MODE == 0) { a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c; a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c; }
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) a0 = a0 * m + c;
        }
    }
    if (clk && (threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = wall_clock64() - w0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE> void run_cu(const char* name, int w, float* d, long long* clk) {
    const int cus = 256, lds = 96 * 1024;
    hipFuncSetAttribute((const void*)kcu<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int warm = 0; warm < 3; ++warm) hipLaunchKernelGGL(kcu<MODE>, dim3(cus), dim3(256 * w), lds, 0, d, 1.f, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kcu<MODE>, dim3(cus), dim3(256 * w), lds, 0, d, 1.f, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[256 * 16];
    hipMemcpy(h, clk, sizeof(long long) * cus * 4 * w, hipMemcpyDeviceToHost);
    double sw = 0, mx = 0;
    for (int i = 0; i < cus * 4 * w; ++i) { sw += (double)h[i]; mx = h[i] > mx ? (double)h[i] : mx; }
    sw /= cus * 4 * w;
    const double insts = 8.0 * ITER;
    printf("[cu] %-28s W = %d waves on every SIMD: kernel %.3f ms (HIP events), per wave %.3f ms mean / %.3f max (s_memrealtime) -> %.3f ns per VALU instruction per SIMD "
           "(= %.2f cycles at 2.4 GHz, %.2f at 2.1 GHz)\n", name, w, ms, sw * 1e-5, mx * 1e-5, sw * 10.0 / (insts * w), sw * 10.0 / (insts * w) * 2.4, sw * 10.0 / (insts * w) * 2.1);
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'c') {
        float* d; hipMalloc(&d, 256 * 1024 * sizeof(float));
        long long* clk; hipMalloc(&clk, sizeof(long long) * 256 * 16);
        for (int w : {1, 2, 3, 4}) { run_cu<0>("8 independent v_fma_f32", w, d, clk); run_cu<1>("dependent v_fma_f32 chain", w, d, clk); }
        return 0;
    }
    float* d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    for (int w : {1, 2, 4, 8}) {
        run<0>("8 independent v_fma_f32", w, d);
        run<1>("dependent v_fma_f32 chain", w, d);
        run<4>("2 interleaved dependent chains", w, d);
        run<2>("8 independent v_pk_fma_f32", w, d);
        run<3>("dependent v_pk_fma_f32 chain", w, d);
    }
    return 0;
}
