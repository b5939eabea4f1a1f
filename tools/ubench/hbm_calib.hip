// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts in the access widths the physics kernel uses
// (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern before trusting an absolute"):
//   rd4 / rd16:  every lane reads 4 / 16 bytes of a 1 GiB buffer once (coalesced), nothing written but one dword per workgroup
//   wr4 / wr16:  every lane writes 4 / 16 bytes of a 1 GiB buffer once
//   wr4s:        4-byte stores at a 52-byte stride (row starts of a [N,13] float tensor: partial lines)
// Build + run on the GPU box: tools/hbm_calib.sh.  The buffers are larger than the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr size_t BYTES = 1ull << 30;

__global__ void rd4(const float* __restrict__ p, float* out, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 123.456f) out[blockIdx.x] = s;
}
__global__ void rd16(const f4* __restrict__ p, float* out, size_t n) {
    f4 s{0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s.x + s.y + s.z + s.w == 123.456f) out[blockIdx.x] = s.x;
}
__global__ void wr4(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)i;
}
__global__ void wr16(f4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = f4{(float)i, 1.f, 2.f, 3.f};
}
__global__ void wr4s(float* p, size_t rows) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) p[i * 13] = (float)i;
}

int main() {
    float *a, *o;
    hipMalloc(&a, BYTES);
    hipMalloc(&o, 1 << 20);
    hipMemset(a, 0, BYTES);
    const dim3 g(256 * 16), b(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(rd4, g, b, 0, 0, a, o, BYTES / 4);
        hipLaunchKernelGGL(rd16, g, b, 0, 0, (const f4*)a, o, BYTES / 16);
        hipLaunchKernelGGL(wr4, g, b, 0, 0, a, BYTES / 4);
        hipLaunchKernelGGL(wr16, g, b, 0, 0, (f4*)a, BYTES / 16);
        hipLaunchKernelGGL(wr4s, g, b, 0, 0, a, BYTES / 52);
    }
    hipDeviceSynchronize();
    printf("known bytes: rd4 rd16 wr4 wr16 = %zu each; wr4s = %zu useful bytes in %zu rows of 52 B\n", BYTES, BYTES / 52 * 4, BYTES / 52);
    return 0;
}
