// Small dense-matrix helpers shared by the physics kernels (registers only: every index is a compile-time constant).
#pragma once
#include "v2p_dev.hpp"

namespace v2p {

struct Sym3 {
    float xx, xy, xz, yy, yz, zz;
};

__device__ __forceinline__ V3 mul(const Sym3& s, V3 v) {
    return V3{s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z, s.xz * v.x + s.yz * v.y + s.zz * v.z};
}
__device__ __forceinline__ V3 row(const M3& m, int i) { return V3{m.m[3 * i], m.m[3 * i + 1], m.m[3 * i + 2]}; }
__device__ __forceinline__ V3 col(const M3& m, int j) { return V3{m.m[j], m.m[3 + j], m.m[6 + j]}; }

// inverse of a symmetric positive definite 3x3
__device__ __forceinline__ Sym3 inv(const Sym3& a) {
    float c00 = a.yy * a.zz - a.yz * a.yz;
    float c01 = a.xz * a.yz - a.xy * a.zz;
    float c02 = a.xy * a.yz - a.xz * a.yy;
    float det = a.xx * c00 + a.xy * c01 + a.xz * c02;
    float id = PHYS_RCP(det);
    Sym3 r;
    r.xx = c00 * id; r.xy = c01 * id; r.xz = c02 * id;
    r.yy = (a.xx * a.zz - a.xz * a.xz) * id;
    r.yz = (a.xy * a.xz - a.xx * a.yz) * id;
    r.zz = (a.xx * a.yy - a.xy * a.xy) * id;
    return r;
}
// Sym3 * M3
__device__ __forceinline__ M3 mul(const Sym3& s, const M3& b) {
    M3 r;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        V3 c = mul(s, col(b, j));
        r.m[j] = c.x; r.m[3 + j] = c.y; r.m[6 + j] = c.z;
    }
    return r;
}

// symmetric 6x6 in 21 floats, packed lower triangle row by row: idx(i,j) = i(i+1)/2 + j, j<=i
__device__ __forceinline__ constexpr int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// inverse of a symmetric positive definite 6x6 (Cholesky, fully unrolled: registers only)
__device__ __forceinline__ void spd6_inverse(const float a[21], float out[21]) {
    float L[21];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float s = a[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= L[tri(j, k)] * L[tri(j, k)];
        float inv_l = rsqrtf(s);
        L[tri(j, j)] = inv_l;  // store 1/l_jj on the diagonal
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            float t = a[tri(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= L[tri(i, k)] * L[tri(j, k)];
            L[tri(i, j)] = t * inv_l;
        }
    }
    // Linv (lower) by forward substitution, column by column
    float Li[21];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int i = c; i < 6; ++i) {
            float s = (i == c) ? 1.f : 0.f;
#pragma unroll
            for (int k = c; k < i; ++k) s -= L[tri(i, k)] * Li[tri(k, c)];
            Li[tri(i, c)] = s * L[tri(i, i)];
        }
    }
    // out = Linv^T Linv
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = i; k < 6; ++k) s += Li[tri(k, i)] * Li[tri(k, j)];
            out[tri(i, j)] = s;
        }
}

// 6x6 symmetric <-> blocks: [A B; B^T C]; packed as A(6) B(9) C(6)
struct Blocks {
    Sym3 A;
    M3 B;
    Sym3 C;
};


}  // namespace v2p
