// Articulated rigid-body step for the SMPL humanoid: the replacement of the two
// `gym.simulate` calls per control step (embodied_pose/env/tasks/base_task.py:450-454; sim
// parameters embodied_pose/cfg/amass_im.yaml:37-52; actor setup humanoid_smpl_im.py:273-276,
// 356-389).  The model ("v2p physics v1") is specified in oracle/phys/v2p_phys_oracle.c,
// which evaluates it with dense matrices in float64; this kernel evaluates the same model in
// float32 with O(n) recursions:
//
//   pass 1  root->leaves  kinematics, velocity-product terms, body inertia/bias (world axes,
//                         referred to each body's own origin), implicit-PD joint torque
//   pass 2  leaves->root  articulated-body inertia with the joint diagonal augmented by
//                         armature + h*kd + h^2*kp  (== solving with Mt = M + diag)
//   pass 3  root->leaves  accelerations -> unconstrained link velocities v*
//   contacts              hull vertices vs plane, <=4 point manifold per body
//   operational-space inverse inertia Lambda_b per body (root->leaves recursion)
//   PGS                   block Gauss-Seidel: per body, rows solved against Lambda_b, the
//                         net impulse propagated through the tree (leaf->root->leaves)
//   integrate             generalized velocities recovered from link velocities
//
// Mapping: ONE ENVIRONMENT PER LANE.  The body model is identical for every lane, so model
// data comes in through scalar loads; per-env data lives in structure-of-arrays buffers
// [slot][env] so that every lane-wide access is one coalesced 256-byte transaction.  Loops
// over links / vertices / rows are wave-uniform (scalar control flow, no divergence except
// value selects).
#include <math.h>
#include <stdlib.h>

#include "v2p_internal.hpp"
#include "v2p_math.hpp"

namespace v2p {

// ---------------------------------------------------------------------------- workspace layout
// per-link slots
constexpr int KQ = 0;     // 4  world quaternion
constexpr int KX = 4;     // 3  origin position
constexpr int KW = 7;     // 3  angular velocity (pass1: old; pass3 on: v*)
constexpr int KV = 10;    // 3  origin linear velocity
constexpr int KZW = 13;   // 3  velocity-product angular acceleration term
constexpr int KZV = 16;   // 3  velocity-product linear acceleration term
constexpr int KT = 19;    // 3  joint torque (world axes)
constexpr int KIA = 22;   // 21 articulated inertia A(6) B(9) C(6); reused for Lambda_b after pass 2
constexpr int KP = 43;    // 6  articulated bias force (n, f)
constexpr int KDI = 49;   // 6  D^-1
constexpr int KE = 55;    // 9  E = D^-1 B
constexpr int KU = 64;    // 3  u
constexpr int KR = 67;    // 3  r = x_b - x_parent
constexpr int KA = 70;    // 6  acceleration (pass 3) / delta-velocity (impulse propagation)
constexpr int KDU = 76;   // 3  delta-u on the impulse path
constexpr int KCN = 79;   // 1  number of contacts
constexpr int KCR = 80;   // 12 contact offsets from the body origin
constexpr int KCB = 92;   // 4  contact bias
constexpr int KCL = 96;   // 12 contact impulses (n, t1, t2)
constexpr int LINK_SLOTS = 108;
constexpr int WS_SLOTS = LINK_SLOTS * NB;

int physics_ws_slots() { return WS_SLOTS; }

struct PhysArgs {
    const DevModel* __restrict__ model;
    float* __restrict__ state;
    const float* __restrict__ ctrl;
    float* __restrict__ out;
    float* __restrict__ ws;
    int32_t* __restrict__ contact_ids;
    int64_t n;
    int lanes;  // environments per wave64 (active lanes); the rest of the wave shadows the last one
    EnvParams p;
};

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
    float id = 1.f / det;
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

// ---------------------------------------------------------------------------- the kernel
#define WSL(b, k) ws[((b) * LINK_SLOTS + (k)) * N + e]

__device__ __forceinline__ V3 ws3(const float* ws, int64_t N, int64_t e, int b, int k) {
    return V3{WSL(b, k), WSL(b, k + 1), WSL(b, k + 2)};
}
__device__ __forceinline__ void ws3s(float* ws, int64_t N, int64_t e, int b, int k, V3 v) {
    WSL(b, k) = v.x; WSL(b, k + 1) = v.y; WSL(b, k + 2) = v.z;
}
__device__ __forceinline__ Sym3 wsS(const float* ws, int64_t N, int64_t e, int b, int k) {
    return Sym3{WSL(b, k), WSL(b, k + 1), WSL(b, k + 2), WSL(b, k + 3), WSL(b, k + 4), WSL(b, k + 5)};
}
__device__ __forceinline__ void wsSs(float* ws, int64_t N, int64_t e, int b, int k, const Sym3& s) {
    WSL(b, k) = s.xx; WSL(b, k + 1) = s.xy; WSL(b, k + 2) = s.xz; WSL(b, k + 3) = s.yy; WSL(b, k + 4) = s.yz; WSL(b, k + 5) = s.zz;
}
__device__ __forceinline__ M3 wsM(const float* ws, int64_t N, int64_t e, int b, int k) {
    M3 m;
#pragma unroll
    for (int i = 0; i < 9; ++i) m.m[i] = WSL(b, k + i);
    return m;
}
__device__ __forceinline__ void wsMs(float* ws, int64_t N, int64_t e, int b, int k, const M3& m) {
#pragma unroll
    for (int i = 0; i < 9; ++i) WSL(b, k + i) = m.m[i];
}

// Lambda (6x6 symmetric, blocks La (w-n), Lb (w-f), Lc (v-f)) stored as A(6) B(9) C(6) in the KIA slots
__device__ __forceinline__ Blocks wsBlocks(const float* ws, int64_t N, int64_t e, int b, int k) {
    Blocks r;
    r.A = wsS(ws, N, e, b, k);
    r.B = wsM(ws, N, e, b, k + 6);
    r.C = wsS(ws, N, e, b, k + 15);
    return r;
}
__device__ __forceinline__ void wsBlocksS(float* ws, int64_t N, int64_t e, int b, int k, const Blocks& r) {
    wsSs(ws, N, e, b, k, r.A);
    wsMs(ws, N, e, b, k + 6, r.B);
    wsSs(ws, N, e, b, k + 15, r.C);
}

template <bool CONTACT>
__global__ __launch_bounds__(64) void physics_kernel(PhysArgs a) {
    const int64_t N = a.n;
    // `lanes` environments per wave: the step is bound by dependent memory round trips, not by VALU
    // work, so spreading the envs over more (partly filled) waves buys latency hiding.
    int lane = threadIdx.x < a.lanes ? threadIdx.x : a.lanes - 1;
    int64_t e = (int64_t)blockIdx.x * a.lanes + lane;
    if (e >= N) e = N - 1;  // tail lanes shadow the last env (identical values, identical addresses)
    const DevModel& M = *a.model;
    float* __restrict__ ws = a.ws;
    float* __restrict__ st = a.state;
    const EnvParams& P = a.p;
    const float h = P.h;

    for (int sub = 0; sub < P.nsub; ++sub) {
        const bool wrench_on = sub < P.hold_sub;
        // ================================================================ pass 1: root -> leaves
        for (int b = 0; b < NB; ++b) {
            const int par = M.parents[b];
            Q4 q;
            V3 x, w, xd, zw{0.f, 0.f, 0.f}, zv{0.f, 0.f, 0.f}, tau{0.f, 0.f, 0.f}, r{0.f, 0.f, 0.f};
            M3 R;
            if (b == 0) {
                q = Q4{st[(ST_ROOT_QUAT + 0) * N + e], st[(ST_ROOT_QUAT + 1) * N + e], st[(ST_ROOT_QUAT + 2) * N + e], st[(ST_ROOT_QUAT + 3) * N + e]};
                x = V3{st[(ST_ROOT_POS + 0) * N + e], st[(ST_ROOT_POS + 1) * N + e], st[(ST_ROOT_POS + 2) * N + e]};
                xd = V3{st[(ST_VEL + 0) * N + e], st[(ST_VEL + 1) * N + e], st[(ST_VEL + 2) * N + e]};
                w = V3{st[(ST_VEL + 3) * N + e], st[(ST_VEL + 4) * N + e], st[(ST_VEL + 5) * N + e]};
                R = q2mat(q);
            } else {
                const int jb = ST_JQUAT + 4 * (b - 1);
                Q4 jq{st[(jb + 0) * N + e], st[(jb + 1) * N + e], st[(jb + 2) * N + e], st[(jb + 3) * N + e]};
                const int vb = ST_VEL + 6 + 3 * (b - 1);
                V3 wt{st[(vb + 0) * N + e], st[(vb + 1) * N + e], st[(vb + 2) * N + e]};  // joint rate, body-b axes
                Q4 qp{WSL(par, KQ), WSL(par, KQ + 1), WSL(par, KQ + 2), WSL(par, KQ + 3)};
                V3 xp = ws3(ws, N, e, par, KX), wp = ws3(ws, N, e, par, KW), xdp = ws3(ws, N, e, par, KV);
                q = qnormalize(qmul(qp, jq));
                R = q2mat(q);
                M3 Rp = q2mat(qp);
                r = mul(Rp, V3{M.local_pos[b][0], M.local_pos[b][1], M.local_pos[b][2]});
                x = xp + r;
                V3 wrel = mul(R, wt);
                w = wp + wrel;
                V3 wpr = cross(wp, r);
                xd = xdp + wpr;
                zw = cross(wp, wrel);
                zv = cross(wp, wpr);
                // implicit PD drive: kp (q_tar - q) - (kd + h kp) wrel, q = exp-map of the joint quaternion
                V3 qe = quat_to_expmap_stable(jq);
                const int cb = CT_PD + 3 * (b - 1);
                V3 tar{a.ctrl[(cb + 0) * N + e], a.ctrl[(cb + 1) * N + e], a.ctrl[(cb + 2) * N + e]};
                float kp = M.kp[b], kdh = M.kd[b] + h * M.kp[b];
                V3 tb = kp * (tar - qe) - kdh * wt;
                tau = mul(R, tb);
            }
            WSL(b, KQ) = q.x; WSL(b, KQ + 1) = q.y; WSL(b, KQ + 2) = q.z; WSL(b, KQ + 3) = q.w;
            ws3s(ws, N, e, b, KX, x);
            ws3s(ws, N, e, b, KW, w);
            ws3s(ws, N, e, b, KV, xd);
            ws3s(ws, N, e, b, KZW, zw);
            ws3s(ws, N, e, b, KZV, zv);
            ws3s(ws, N, e, b, KT, tau);
            ws3s(ws, N, e, b, KR, r);
            // body inertia at its origin, world axes
            const float m = M.mass[b];
            V3 d = mul(R, V3{M.com[b][0], M.com[b][1], M.com[b][2]});
            Sym3 Ib{M.inertia[b][0], M.inertia[b][1], M.inertia[b][2], M.inertia[b][3], M.inertia[b][4], M.inertia[b][5]};
            // Ic = R Ib R^T
            V3 c0 = mul(Ib, V3{R.m[0], R.m[1], R.m[2]});  // Ib * (row 0 of R)^T
            V3 c1 = mul(Ib, V3{R.m[3], R.m[4], R.m[5]});
            V3 c2 = mul(Ib, V3{R.m[6], R.m[7], R.m[8]});
            V3 r0 = row(R, 0), r1 = row(R, 1), r2 = row(R, 2);
            Sym3 Ic{dot(r0, c0), dot(r0, c1), dot(r0, c2), dot(r1, c1), dot(r1, c2), dot(r2, c2)};
            float dd = dot(d, d);
            Sym3 A{Ic.xx + m * (dd - d.x * d.x), Ic.xy - m * d.x * d.y, Ic.xz - m * d.x * d.z,
                   Ic.yy + m * (dd - d.y * d.y), Ic.yz - m * d.y * d.z, Ic.zz + m * (dd - d.z * d.z)};
            wsSs(ws, N, e, b, KIA, A);
            // B = m [d]x
            M3 B;
            B.m[0] = 0.f;       B.m[1] = -m * d.z;  B.m[2] = m * d.y;
            B.m[3] = m * d.z;   B.m[4] = 0.f;       B.m[5] = -m * d.x;
            B.m[6] = -m * d.y;  B.m[7] = m * d.x;   B.m[8] = 0.f;
            wsMs(ws, N, e, b, KIA + 6, B);
            wsSs(ws, N, e, b, KIA + 15, Sym3{m, 0.f, 0.f, m, 0.f, m});
            // bias force (velocity terms - gravity - external wrench)
            V3 wwd = cross(w, cross(w, d));
            V3 fl = m * (wwd - V3{0.f, 0.f, P.gravity_z});
            V3 nn = cross(w, mul(Ic, w)) + cross(d, fl);
            if (b == 0 && wrench_on) {
                V3 F{a.ctrl[(CT_FORCE + 0) * N + e], a.ctrl[(CT_FORCE + 1) * N + e], a.ctrl[(CT_FORCE + 2) * N + e]};
                V3 T{a.ctrl[(CT_TORQUE + 0) * N + e], a.ctrl[(CT_TORQUE + 1) * N + e], a.ctrl[(CT_TORQUE + 2) * N + e]};
                nn = nn - T - cross(d, F);  // force acts at the root COM
                fl = fl - F;
            }
            ws3s(ws, N, e, b, KP, nn);
            ws3s(ws, N, e, b, KP + 3, fl);
        }

        // ================================================================ pass 2: leaves -> root
        for (int b = NB - 1; b >= 1; --b) {
            const int par = M.parents[b];
            const float aug = M.arm[b] + h * M.kd[b] + h * h * M.kp[b];
            Sym3 A = wsS(ws, N, e, b, KIA);
            M3 B = wsM(ws, N, e, b, KIA + 6);
            Sym3 C = wsS(ws, N, e, b, KIA + 15);
            V3 pn = ws3(ws, N, e, b, KP), pf = ws3(ws, N, e, b, KP + 3);
            V3 zw = ws3(ws, N, e, b, KZW), zv = ws3(ws, N, e, b, KZV);
            V3 tau = ws3(ws, N, e, b, KT);
            V3 r = ws3(ws, N, e, b, KR);
            Sym3 D{A.xx + aug, A.xy, A.xz, A.yy + aug, A.yz, A.zz + aug};
            Sym3 Di = inv(D);
            M3 E = mul(Di, B);
            V3 u = tau - pn;
            wsSs(ws, N, e, b, KDI, Di);
            wsMs(ws, N, e, b, KE, E);
            ws3s(ws, N, e, b, KU, u);
            // articulated inertia seen through the joint
            Sym3 Aa{aug * (1.f - aug * Di.xx), -aug * aug * Di.xy, -aug * aug * Di.xz, aug * (1.f - aug * Di.yy), -aug * aug * Di.yz,
                    aug * (1.f - aug * Di.zz)};
            M3 Ba;
#pragma unroll
            for (int i = 0; i < 9; ++i) Ba.m[i] = aug * E.m[i];
            V3 b0 = col(B, 0), b1 = col(B, 1), b2 = col(B, 2), e0 = col(E, 0), e1 = col(E, 1), e2 = col(E, 2);
            Sym3 Ca{C.xx - dot(b0, e0), C.xy - dot(b0, e1), C.xz - dot(b0, e2), C.yy - dot(b1, e1), C.yz - dot(b1, e2), C.zz - dot(b2, e2)};
            V3 Diu = mul(Di, u);
            V3 pan = pn + mul(Aa, zw) + mul(Ba, zv) + (u - aug * Diu);
            V3 paf = pf + V3{dot(col(Ba, 0), zw), dot(col(Ba, 1), zw), dot(col(Ba, 2), zw)} + mul(Ca, zv) +
                     V3{dot(e0, u), dot(e1, u), dot(e2, u)};
            // shift to the parent origin and accumulate
            // S = [r]x Ca (columns r x Ca_col); Ca symmetric so Ca_col j = row j
            V3 s0 = cross(r, V3{Ca.xx, Ca.xy, Ca.xz}), s1 = cross(r, V3{Ca.xy, Ca.yy, Ca.yz}), s2 = cross(r, V3{Ca.xz, Ca.yz, Ca.zz});
            M3 Y;  // parent B increment = Ba + S
            Y.m[0] = Ba.m[0] + s0.x; Y.m[1] = Ba.m[1] + s1.x; Y.m[2] = Ba.m[2] + s2.x;
            Y.m[3] = Ba.m[3] + s0.y; Y.m[4] = Ba.m[4] + s1.y; Y.m[5] = Ba.m[5] + s2.y;
            Y.m[6] = Ba.m[6] + s0.z; Y.m[7] = Ba.m[7] + s1.z; Y.m[8] = Ba.m[8] + s2.z;
            // T1 rows = r x Ba_row ; T2 rows = r x S_row
            V3 t10 = cross(r, row(Ba, 0)), t11 = cross(r, row(Ba, 1)), t12 = cross(r, row(Ba, 2));
            V3 sr0{s0.x, s1.x, s2.x}, sr1{s0.y, s1.y, s2.y}, sr2{s0.z, s1.z, s2.z};
            V3 t20 = cross(r, sr0), t21 = cross(r, sr1), t22 = cross(r, sr2);
            Sym3 Ainc{Aa.xx + 2.f * t10.x + t20.x, Aa.xy + t10.y + t11.x + t20.y, Aa.xz + t10.z + t12.x + t20.z,
                      Aa.yy + 2.f * t11.y + t21.y, Aa.yz + t11.z + t12.y + t21.z, Aa.zz + 2.f * t12.z + t22.z};
            V3 pinc_n = pan + cross(r, paf);
            // read-modify-write of the parent's accumulators
            WSL(par, KIA + 0) += Ainc.xx; WSL(par, KIA + 1) += Ainc.xy; WSL(par, KIA + 2) += Ainc.xz;
            WSL(par, KIA + 3) += Ainc.yy; WSL(par, KIA + 4) += Ainc.yz; WSL(par, KIA + 5) += Ainc.zz;
#pragma unroll
            for (int i = 0; i < 9; ++i) WSL(par, KIA + 6 + i) += Y.m[i];
            WSL(par, KIA + 15) += Ca.xx; WSL(par, KIA + 16) += Ca.xy; WSL(par, KIA + 17) += Ca.xz;
            WSL(par, KIA + 18) += Ca.yy; WSL(par, KIA + 19) += Ca.yz; WSL(par, KIA + 20) += Ca.zz;
            WSL(par, KP + 0) += pinc_n.x; WSL(par, KP + 1) += pinc_n.y; WSL(par, KP + 2) += pinc_n.z;
            WSL(par, KP + 3) += paf.x; WSL(par, KP + 4) += paf.y; WSL(par, KP + 5) += paf.z;
        }

        // ================================================================ root: 6x6 solve
        {
            Blocks I0 = wsBlocks(ws, N, e, 0, KIA);
            float a6[21], inv6[21];
            // rows/cols 0-2 = angular (A), 3-5 = linear (C), off-diagonal block (row 3+i, col j) = B^T(i,j) = B(j,i)
            a6[tri(0, 0)] = I0.A.xx; a6[tri(1, 0)] = I0.A.xy; a6[tri(2, 0)] = I0.A.xz; a6[tri(1, 1)] = I0.A.yy; a6[tri(2, 1)] = I0.A.yz; a6[tri(2, 2)] = I0.A.zz;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) a6[tri(3 + i, j)] = I0.B.m[3 * j + i];
            a6[tri(3, 3)] = I0.C.xx; a6[tri(4, 3)] = I0.C.xy; a6[tri(5, 3)] = I0.C.xz; a6[tri(4, 4)] = I0.C.yy; a6[tri(5, 4)] = I0.C.yz; a6[tri(5, 5)] = I0.C.zz;
            spd6_inverse(a6, inv6);
            Blocks L0;
            L0.A = Sym3{inv6[tri(0, 0)], inv6[tri(1, 0)], inv6[tri(2, 0)], inv6[tri(1, 1)], inv6[tri(2, 1)], inv6[tri(2, 2)]};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) L0.B.m[3 * i + j] = inv6[tri(3 + j, i)];  // Lb(i,j): w_i response to f_j
            L0.C = Sym3{inv6[tri(3, 3)], inv6[tri(4, 3)], inv6[tri(5, 3)], inv6[tri(4, 4)], inv6[tri(5, 4)], inv6[tri(5, 5)]};
            wsBlocksS(ws, N, e, 0, KIA, L0);  // Lambda_0
            V3 pn = ws3(ws, N, e, 0, KP), pf = ws3(ws, N, e, 0, KP + 3);
            // a0 = -Lambda_0 p
            V3 al = -(mul(L0.A, pn) + mul(L0.B, pf));
            V3 ac = -(V3{dot(col(L0.B, 0), pn), dot(col(L0.B, 1), pn), dot(col(L0.B, 2), pn)} + mul(L0.C, pf));
            ws3s(ws, N, e, 0, KA, al);
            ws3s(ws, N, e, 0, KA + 3, ac);
            V3 w = ws3(ws, N, e, 0, KW) + h * al;
            V3 xd = ws3(ws, N, e, 0, KV) + h * ac;
            ws3s(ws, N, e, 0, KW, w);
            ws3s(ws, N, e, 0, KV, xd);
        }

        // ================================================================ pass 3: root -> leaves (accelerations, v*)
        for (int b = 1; b < NB; ++b) {
            const int par = M.parents[b];
            const float aug = M.arm[b] + h * M.kd[b] + h * h * M.kp[b];
            V3 alp = ws3(ws, N, e, par, KA), acp = ws3(ws, N, e, par, KA + 3);
            V3 r = ws3(ws, N, e, b, KR);
            V3 aw = alp + ws3(ws, N, e, b, KZW);
            V3 av = acp + cross(alp, r) + ws3(ws, N, e, b, KZV);
            Sym3 Di = wsS(ws, N, e, b, KDI);
            M3 E = wsM(ws, N, e, b, KE);
            V3 u = ws3(ws, N, e, b, KU);
            V3 qdd = mul(Di, u + aug * aw) - aw - mul(E, av);
            V3 al = aw + qdd;
            ws3s(ws, N, e, b, KA, al);
            ws3s(ws, N, e, b, KA + 3, av);
            // unconstrained link velocity at the OLD configuration: w_b* = w_p* + (wrel + h qdd), xd_b* = xd_p* + w_p* x r
            // old relative angular velocity (world axes): w_p(old) is already overwritten, so rebuild it from the joint rate
            M3 R = q2mat(Q4{WSL(b, KQ), WSL(b, KQ + 1), WSL(b, KQ + 2), WSL(b, KQ + 3)});
            const int vb = ST_VEL + 6 + 3 * (b - 1);
            V3 wt{st[(vb + 0) * N + e], st[(vb + 1) * N + e], st[(vb + 2) * N + e]};
            V3 wnew = ws3(ws, N, e, par, KW) + mul(R, wt) + h * qdd;
            V3 xdnew = ws3(ws, N, e, par, KV) + cross(ws3(ws, N, e, par, KW), r);
            ws3s(ws, N, e, b, KW, wnew);
            ws3s(ws, N, e, b, KV, xdnew);
        }

        if (CONTACT) {
            // ============================================================ contact generation
            const float coff = P.contact_offset;
            for (int b = 0; b < NB; ++b) {
                V3 x = ws3(ws, N, e, b, KX);
                int cnt = 0;
                bool near = x.z - M.bound_radius[b] < coff;
                if (__any(near)) {
                    M3 R = q2mat(Q4{WSL(b, KQ), WSL(b, KQ + 1), WSL(b, KQ + 2), WSL(b, KQ + 3)});
                    const int v0 = M.hull_offsets[b], v1 = M.hull_offsets[b + 1];
                    int f0 = -1, f1 = -1, f2 = -1, f3 = -1, k0 = -1;
                    float zmin = 0.f;
                    for (int v = v0; v < v1; ++v) {
                        float z = x.z + R.m[6] * M.hull_verts[v][0] + R.m[7] * M.hull_verts[v][1] + R.m[8] * M.hull_verts[v][2];
                        bool c = z < coff;
                        int i = v - v0;
                        f0 = (c && cnt == 0) ? i : f0;
                        f1 = (c && cnt == 1) ? i : f1;
                        f2 = (c && cnt == 2) ? i : f2;
                        f3 = (c && cnt == 3) ? i : f3;
                        bool better = c && (k0 < 0 || z < zmin);
                        k0 = better ? i : k0;
                        zmin = better ? z : zmin;
                        cnt += c ? 1 : 0;
                    }
                    int s0 = f0, s1 = f1, s2 = f2, s3 = f3, ns = cnt < 4 ? cnt : 4;
                    if (__any(cnt > 4)) {
                        // manifold reduction: deepest, farthest from it, extreme on either side of that line
                        int kk0 = k0 < 0 ? 0 : k0;
                        V3 u0{M.hull_verts[v0 + kk0][0], M.hull_verts[v0 + kk0][1], M.hull_verts[v0 + kk0][2]};
                        float p0x = x.x + R.m[0] * u0.x + R.m[1] * u0.y + R.m[2] * u0.z;
                        float p0y = x.y + R.m[3] * u0.x + R.m[4] * u0.y + R.m[5] * u0.z;
                        int k1 = -1;
                        float best = -1.f;
                        for (int v = v0; v < v1; ++v) {
                            float ux = M.hull_verts[v][0], uy = M.hull_verts[v][1], uz = M.hull_verts[v][2];
                            float z = x.z + R.m[6] * ux + R.m[7] * uy + R.m[8] * uz;
                            float dx = x.x + R.m[0] * ux + R.m[1] * uy + R.m[2] * uz - p0x;
                            float dy = x.y + R.m[3] * ux + R.m[4] * uy + R.m[5] * uz - p0y;
                            float d2 = dx * dx + dy * dy;
                            int i = v - v0;
                            bool take = (z < coff) && (i != k0) && (d2 > best);
                            best = take ? d2 : best;
                            k1 = take ? i : k1;
                        }
                        int kk1 = k1 < 0 ? 0 : k1;
                        V3 u1{M.hull_verts[v0 + kk1][0], M.hull_verts[v0 + kk1][1], M.hull_verts[v0 + kk1][2]};
                        float ex = x.x + R.m[0] * u1.x + R.m[1] * u1.y + R.m[2] * u1.z - p0x;
                        float ey = x.y + R.m[3] * u1.x + R.m[4] * u1.y + R.m[5] * u1.z - p0y;
                        int k2 = -1, k3 = -1;
                        float amax = 0.f, amin = 0.f;
                        for (int v = v0; v < v1; ++v) {
                            float ux = M.hull_verts[v][0], uy = M.hull_verts[v][1], uz = M.hull_verts[v][2];
                            float z = x.z + R.m[6] * ux + R.m[7] * uy + R.m[8] * uz;
                            float dx = x.x + R.m[0] * ux + R.m[1] * uy + R.m[2] * uz - p0x;
                            float dy = x.y + R.m[3] * ux + R.m[4] * uy + R.m[5] * uz - p0y;
                            float area = ex * dy - ey * dx;
                            int i = v - v0;
                            bool cand = (z < coff) && (i != k0) && (i != k1);
                            bool up = cand && area > amax;
                            bool dn = cand && area < amin;
                            amax = up ? area : amax; k2 = up ? i : k2;
                            amin = dn ? area : amin; k3 = dn ? i : k3;
                        }
                        if (cnt > 4) {
                            s0 = k0; s1 = k1; ns = 2;
                            s2 = k2 >= 0 ? k2 : k3;
                            s3 = k2 >= 0 ? k3 : -1;
                            ns += (k2 >= 0 ? 1 : 0) + (k3 >= 0 ? 1 : 0);
                        }
                    }
                    cnt = ns;
                    int sel[4] = {s0, s1, s2, s3};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        int vi = sel[c] < 0 ? 0 : sel[c];
                        const float* uv = &M.hull_verts[v0 + vi][0];
                        V3 rr = mul(R, V3{uv[0], uv[1], uv[2]});
                        ws3s(ws, N, e, b, KCR + 3 * c, rr);
                        float d = x.z + rr.z;
                        WSL(b, KCB + c) = d >= 0.f ? d / h : fmaxf(P.erp * d / h, -P.max_depen);
                        WSL(b, KCL + 3 * c) = 0.f; WSL(b, KCL + 3 * c + 1) = 0.f; WSL(b, KCL + 3 * c + 2) = 0.f;
                        a.contact_ids[(e * NB + b) * 4 + c] = c < cnt ? b * 64 + sel[c] : -1;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) a.contact_ids[(e * NB + b) * 4 + c] = -1;
                }
                WSL(b, KCN) = (float)cnt;
            }

            // ============================================================ Lambda_b recursion (root -> leaves)
            for (int b = 1; b < NB; ++b) {
                const int par = M.parents[b];
                const float aug = M.arm[b] + h * M.kd[b] + h * h * M.kp[b];
                Blocks Lp = wsBlocks(ws, N, e, par, KIA);
                V3 r = ws3(ws, N, e, b, KR);
                Sym3 Di = wsS(ws, N, e, b, KDI);
                M3 E = wsM(ws, N, e, b, KE);
                // G = X Lp X^T: Ga = La ; Gb = La [r]x + Lb ; Gc = Lc - [r]x Lb + Lb^T [r]x - [r]x La [r]x
                // (La [r]x) row i = La_row_i x r
                V3 la0{Lp.A.xx, Lp.A.xy, Lp.A.xz}, la1{Lp.A.xy, Lp.A.yy, Lp.A.yz}, la2{Lp.A.xz, Lp.A.yz, Lp.A.zz};
                M3 Gb;
                {
                    V3 g0 = cross(la0, r) + row(Lp.B, 0), g1 = cross(la1, r) + row(Lp.B, 1), g2 = cross(la2, r) + row(Lp.B, 2);
                    Gb.m[0] = g0.x; Gb.m[1] = g0.y; Gb.m[2] = g0.z; Gb.m[3] = g1.x; Gb.m[4] = g1.y; Gb.m[5] = g1.z; Gb.m[6] = g2.x; Gb.m[7] = g2.y; Gb.m[8] = g2.z;
                }
                // Gc = Lc + ( -[r]x Gb_partial ... ) : use Gc = Lc - [r]x Lb + (Gb^T [r]x) where Gb = La[r]x + Lb:
                //   Gb^T [r]x = [r]x^T... expand: (La[r]x + Lb)^T [r]x = [r]x^T La [r]x + Lb^T [r]x = -[r]x La [r]x + Lb^T [r]x  (matches)
                // (Gb^T [r]x) row i = (Gb^T)_row_i x r = Gb_col_i x r ; (-[r]x Lb) column j = -(r x Lb_col_j)
                Sym3 Gc;
                {
                    V3 q0 = cross(col(Gb, 0), r), q1 = cross(col(Gb, 1), r), q2 = cross(col(Gb, 2), r);  // rows of Gb^T [r]x
                    V3 m0 = cross(r, col(Lp.B, 0)), m1 = cross(r, col(Lp.B, 1)), m2 = cross(r, col(Lp.B, 2));  // columns of [r]x Lb
                    Gc.xx = Lp.C.xx + q0.x - m0.x;
                    Gc.xy = Lp.C.xy + q0.y - m1.x;
                    Gc.xz = Lp.C.xz + q0.z - m2.x;
                    Gc.yy = Lp.C.yy + q1.y - m1.y;
                    Gc.yz = Lp.C.yz + q1.z - m2.y;
                    Gc.zz = Lp.C.zz + q2.z - m2.z;
                }
                // Lambda_b = [Di 0; 0 0] + T^T G T,  T = [aug Di, 0; -E^T, 1]
                // H1 = aug Ga Di - Gb E^T   (3x3);  H2 = aug Gb^T Di - Gc E^T (3x3)
                // La' = Di + aug Di H1 - E H2 ; Lb' = aug Di Gb - E Gc ; Lc' = Gc
                M3 DiM;  // Di as full matrix
                DiM.m[0] = Di.xx; DiM.m[1] = Di.xy; DiM.m[2] = Di.xz; DiM.m[3] = Di.xy; DiM.m[4] = Di.yy; DiM.m[5] = Di.yz; DiM.m[6] = Di.xz; DiM.m[7] = Di.yz; DiM.m[8] = Di.zz;
                M3 H1, H2;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    V3 gai = i == 0 ? la0 : (i == 1 ? la1 : la2);
                    V3 gbi = row(Gb, i);
                    V3 gbti = col(Gb, i);
                    V3 gci = i == 0 ? V3{Gc.xx, Gc.xy, Gc.xz} : (i == 1 ? V3{Gc.xy, Gc.yy, Gc.yz} : V3{Gc.xz, Gc.yz, Gc.zz});
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        V3 dj = col(DiM, j);
                        V3 ej = row(E, j);  // (E^T) column j = E row j
                        H1.m[3 * i + j] = aug * dot(gai, dj) - dot(gbi, ej);
                        H2.m[3 * i + j] = aug * dot(gbti, dj) - dot(gci, ej);
                    }
                }
                Blocks Lb;
                {
                    M3 t;  // aug Di H1 - E H2
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            t.m[3 * i + j] = aug * dot(row(DiM, i), col(H1, j)) - dot(row(E, i), col(H2, j));
                    Lb.A = Sym3{Di.xx + t.m[0], Di.xy + 0.5f * (t.m[1] + t.m[3]), Di.xz + 0.5f * (t.m[2] + t.m[6]), Di.yy + t.m[4],
                                Di.yz + 0.5f * (t.m[5] + t.m[7]), Di.zz + t.m[8]};
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            V3 gcj = j == 0 ? V3{Gc.xx, Gc.xy, Gc.xz} : (j == 1 ? V3{Gc.xy, Gc.yy, Gc.yz} : V3{Gc.xz, Gc.yz, Gc.zz});
                            Lb.B.m[3 * i + j] = aug * dot(row(DiM, i), col(Gb, j)) - dot(row(E, i), gcj);
                        }
                    Lb.C = Gc;
                }
                wsBlocksS(ws, N, e, b, KIA, Lb);
            }

            // ============================================================ block Gauss-Seidel
            for (int it = 0; it < P.n_iter; ++it) {
                for (int b = 0; b < NB; ++b) {
                    int cnt = (int)WSL(b, KCN);
                    if (!__any(cnt > 0)) continue;
                    Blocks L = wsBlocks(ws, N, e, b, KIA);
                    V3 w = ws3(ws, N, e, b, KW), xd = ws3(ws, N, e, b, KV);
                    V3 phin{0.f, 0.f, 0.f}, phif{0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool active = c < cnt;
                        V3 rr = ws3(ws, N, e, b, KCR + 3 * c);
                        float bias = WSL(b, KCB + c);
                        float ln = WSL(b, KCL + 3 * c), l1 = WSL(b, KCL + 3 * c + 1), l2 = WSL(b, KCL + 3 * c + 2);
#pragma unroll
                        for (int ax = 0; ax < 3; ++ax) {
                            V3 dir = ax == 0 ? V3{0.f, 0.f, 1.f} : (ax == 1 ? V3{1.f, 0.f, 0.f} : V3{0.f, 1.f, 0.f});
                            V3 jn = cross(rr, dir);  // row = [jn ; dir] against (w, xd)
                            // y = Lambda [jn; dir]
                            V3 yw = mul(L.A, jn) + mul(L.B, dir);
                            V3 yv = V3{dot(col(L.B, 0), jn), dot(col(L.B, 1), jn), dot(col(L.B, 2), jn)} + mul(L.C, dir);
                            float wii = dot(jn, yw) + dot(dir, yv);
                            float rel = dot(jn, w) + dot(dir, xd) + (ax == 0 ? bias : 0.f);
                            float old = ax == 0 ? ln : (ax == 1 ? l1 : l2);
                            float nl = old - rel / wii;
                            if (ax == 0) nl = fmaxf(nl, 0.f);
                            else { float lim = P.mu * ln; nl = fminf(fmaxf(nl, -lim), lim); }
                            float dl = active ? nl - old : 0.f;
                            if (ax == 0) ln += dl; else if (ax == 1) l1 += dl; else l2 += dl;
                            w = w + dl * yw;
                            xd = xd + dl * yv;
                            phin = phin + dl * jn;
                            phif = phif + dl * dir;
                        }
                        WSL(b, KCL + 3 * c) = ln; WSL(b, KCL + 3 * c + 1) = l1; WSL(b, KCL + 3 * c + 2) = l2;
                    }
                    // ---- propagate the net impulse (phin, phif) applied at link b: leaf -> root
                    V3 nI = phin, fI = phif;
                    for (int i = b; i != 0; i = M.parents[i]) {
                        const float aug = M.arm[i] + h * M.kd[i] + h * h * M.kp[i];
                        ws3s(ws, N, e, i, KDU, nI);
                        Sym3 Di = wsS(ws, N, e, i, KDI);
                        M3 E = wsM(ws, N, e, i, KE);
                        V3 r = ws3(ws, N, e, i, KR);
                        V3 na = aug * mul(Di, nI);
                        V3 fa = fI - V3{dot(col(E, 0), nI), dot(col(E, 1), nI), dot(col(E, 2), nI)};
                        nI = na + cross(r, fa);
                        fI = fa;
                    }
                    // root response
                    {
                        Blocks L0 = wsBlocks(ws, N, e, 0, KIA);
                        V3 dw = mul(L0.A, nI) + mul(L0.B, fI);
                        V3 dv = V3{dot(col(L0.B, 0), nI), dot(col(L0.B, 1), nI), dot(col(L0.B, 2), nI)} + mul(L0.C, fI);
                        ws3s(ws, N, e, 0, KA, dw);
                        ws3s(ws, N, e, 0, KA + 3, dv);
                        ws3s(ws, N, e, 0, KW, ws3(ws, N, e, 0, KW) + dw);
                        ws3s(ws, N, e, 0, KV, ws3(ws, N, e, 0, KV) + dv);
                    }
                    // root -> leaves: every link moves
                    // ancestors-or-self of b (bit mask, wave-uniform)
                    unsigned path = 0;
                    for (int i = b; i != 0; i = M.parents[i]) path |= 1u << i;
                    for (int i = 1; i < NB; ++i) {
                        const int par = M.parents[i];
                        const float aug = M.arm[i] + h * M.kd[i] + h * h * M.kp[i];
                        V3 dwp = ws3(ws, N, e, par, KA), dvp = ws3(ws, N, e, par, KA + 3);
                        V3 r = ws3(ws, N, e, i, KR);
                        Sym3 Di = wsS(ws, N, e, i, KDI);
                        M3 E = wsM(ws, N, e, i, KE);
                        V3 av = dvp + cross(dwp, r);
                        V3 nu = aug * dwp;
                        if ((path >> i) & 1u) nu = nu + ws3(ws, N, e, i, KDU);
                        V3 dw = mul(Di, nu) - mul(E, av);
                        ws3s(ws, N, e, i, KA, dw);
                        ws3s(ws, N, e, i, KA + 3, av);
                        ws3s(ws, N, e, i, KW, ws3(ws, N, e, i, KW) + dw);
                        ws3s(ws, N, e, i, KV, ws3(ws, N, e, i, KV) + av);
                    }
                }
            }
        }

        // ================================================================ velocities -> generalized, damping, clamp, integrate
        const float sc = 1.f / (1.f + h * P.ang_damp);
        const float wmax = P.max_ang_vel;
        {
            V3 w0 = sc * ws3(ws, N, e, 0, KW);
            V3 xd0 = ws3(ws, N, e, 0, KV);
            float n2 = dot(w0, w0);
            if (n2 > wmax * wmax) w0 = (wmax * rsqrtf(n2)) * w0;
            st[(ST_VEL + 0) * N + e] = xd0.x; st[(ST_VEL + 1) * N + e] = xd0.y; st[(ST_VEL + 2) * N + e] = xd0.z;
            st[(ST_VEL + 3) * N + e] = w0.x; st[(ST_VEL + 4) * N + e] = w0.y; st[(ST_VEL + 5) * N + e] = w0.z;
            V3 x0 = ws3(ws, N, e, 0, KX) + h * xd0;
            st[(ST_ROOT_POS + 0) * N + e] = x0.x; st[(ST_ROOT_POS + 1) * N + e] = x0.y; st[(ST_ROOT_POS + 2) * N + e] = x0.z;
            Q4 q0{WSL(0, KQ), WSL(0, KQ + 1), WSL(0, KQ + 2), WSL(0, KQ + 3)};
            Q4 nq = qnormalize(qmul(rotvec_to_quat(h * w0), q0));  // world-frame rate: left multiply
            st[(ST_ROOT_QUAT + 0) * N + e] = nq.x; st[(ST_ROOT_QUAT + 1) * N + e] = nq.y; st[(ST_ROOT_QUAT + 2) * N + e] = nq.z; st[(ST_ROOT_QUAT + 3) * N + e] = nq.w;
        }
        const bool last = sub == P.nsub - 1;
        for (int b = 1; b < NB; ++b) {
            const int par = M.parents[b];
            M3 R = q2mat(Q4{WSL(b, KQ), WSL(b, KQ + 1), WSL(b, KQ + 2), WSL(b, KQ + 3)});
            V3 wt = mulT(R, ws3(ws, N, e, b, KW) - ws3(ws, N, e, par, KW));  // joint rate, body axes (undamped)
            const int jb = ST_JQUAT + 4 * (b - 1);
            Q4 jq{st[(jb + 0) * N + e], st[(jb + 1) * N + e], st[(jb + 2) * N + e], st[(jb + 3) * N + e]};
            if (last) {
                // joint drive torque actually applied over the substep (implicit form)
                V3 qe = quat_to_expmap_stable(jq);
                const int cb = CT_PD + 3 * (b - 1);
                V3 tar{a.ctrl[(cb + 0) * N + e], a.ctrl[(cb + 1) * N + e], a.ctrl[(cb + 2) * N + e]};
                V3 tf = M.kp[b] * (tar - qe - h * wt) - M.kd[b] * wt;
                const int ob = OUT_DOF_FORCE + 3 * (b - 1);
                a.out[(ob + 0) * N + e] = tf.x; a.out[(ob + 1) * N + e] = tf.y; a.out[(ob + 2) * N + e] = tf.z;
            }
            wt = sc * wt;
            float n2 = dot(wt, wt);
            if (n2 > wmax * wmax) wt = (wmax * rsqrtf(n2)) * wt;
            const int vb = ST_VEL + 6 + 3 * (b - 1);
            st[(vb + 0) * N + e] = wt.x; st[(vb + 1) * N + e] = wt.y; st[(vb + 2) * N + e] = wt.z;
            Q4 nq = qnormalize(qmul(jq, rotvec_to_quat(h * wt)));  // body-frame rate: right multiply
            st[(jb + 0) * N + e] = nq.x; st[(jb + 1) * N + e] = nq.y; st[(jb + 2) * N + e] = nq.z; st[(jb + 3) * N + e] = nq.w;
        }
        if (CONTACT && last) {
            // net contact force per body = sum of impulses / h  (refresh_net_contact_force_tensor)
            const float ih = 1.f / h;
            for (int b = 0; b < NB; ++b) {
                int cnt = (int)WSL(b, KCN);
                V3 f{0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < cnt) { f.z += WSL(b, KCL + 3 * c); f.x += WSL(b, KCL + 3 * c + 1); f.y += WSL(b, KCL + 3 * c + 2); }
                a.out[(OUT_CONTACT + 3 * b + 0) * N + e] = f.x * ih;
                a.out[(OUT_CONTACT + 3 * b + 1) * N + e] = f.y * ih;
                a.out[(OUT_CONTACT + 3 * b + 2) * N + e] = f.z * ih;
            }
        }
    }

    // ==================================================================== final kinematics -> rigid-body state, dof_pos
    for (int b = 0; b < NB; ++b) {
        const int par = M.parents[b];
        Q4 q;
        V3 x, w, xd;
        if (b == 0) {
            q = Q4{st[(ST_ROOT_QUAT + 0) * N + e], st[(ST_ROOT_QUAT + 1) * N + e], st[(ST_ROOT_QUAT + 2) * N + e], st[(ST_ROOT_QUAT + 3) * N + e]};
            x = V3{st[(ST_ROOT_POS + 0) * N + e], st[(ST_ROOT_POS + 1) * N + e], st[(ST_ROOT_POS + 2) * N + e]};
            xd = V3{st[(ST_VEL + 0) * N + e], st[(ST_VEL + 1) * N + e], st[(ST_VEL + 2) * N + e]};
            w = V3{st[(ST_VEL + 3) * N + e], st[(ST_VEL + 4) * N + e], st[(ST_VEL + 5) * N + e]};
        } else {
            const int jb = ST_JQUAT + 4 * (b - 1);
            Q4 jq{st[(jb + 0) * N + e], st[(jb + 1) * N + e], st[(jb + 2) * N + e], st[(jb + 3) * N + e]};
            const int vb = ST_VEL + 6 + 3 * (b - 1);
            V3 wt{st[(vb + 0) * N + e], st[(vb + 1) * N + e], st[(vb + 2) * N + e]};
            Q4 qp{WSL(par, KQ), WSL(par, KQ + 1), WSL(par, KQ + 2), WSL(par, KQ + 3)};
            V3 xp = ws3(ws, N, e, par, KX), wp = ws3(ws, N, e, par, KW), xdp = ws3(ws, N, e, par, KV);
            q = qnormalize(qmul(qp, jq));
            V3 r = mul(q2mat(qp), V3{M.local_pos[b][0], M.local_pos[b][1], M.local_pos[b][2]});
            x = xp + r;
            w = wp + mul(q2mat(q), wt);
            xd = xdp + cross(wp, r);
            V3 qe = quat_to_expmap_stable(jq);
            const int ob = OUT_DOF_POS + 3 * (b - 1);
            a.out[(ob + 0) * N + e] = qe.x; a.out[(ob + 1) * N + e] = qe.y; a.out[(ob + 2) * N + e] = qe.z;
        }
        WSL(b, KQ) = q.x; WSL(b, KQ + 1) = q.y; WSL(b, KQ + 2) = q.z; WSL(b, KQ + 3) = q.w;
        ws3s(ws, N, e, b, KX, x);
        ws3s(ws, N, e, b, KW, w);
        ws3s(ws, N, e, b, KV, xd);
        const int ob = OUT_RB + 13 * b;
        a.out[(ob + 0) * N + e] = x.x; a.out[(ob + 1) * N + e] = x.y; a.out[(ob + 2) * N + e] = x.z;
        a.out[(ob + 3) * N + e] = q.x; a.out[(ob + 4) * N + e] = q.y; a.out[(ob + 5) * N + e] = q.z; a.out[(ob + 6) * N + e] = q.w;
        a.out[(ob + 7) * N + e] = xd.x; a.out[(ob + 8) * N + e] = xd.y; a.out[(ob + 9) * N + e] = xd.z;
        a.out[(ob + 10) * N + e] = w.x; a.out[(ob + 11) * N + e] = w.y; a.out[(ob + 12) * N + e] = w.z;
    }
    if (!CONTACT) {
        for (int k = 0; k < NB * 3; ++k) a.out[(OUT_CONTACT + k) * N + e] = 0.f;
    }
}

int launch_env_physics(v2p_env* env, hipStream_t s) {
    PhysArgs a;
    a.model = env->model->dev;
    a.state = env->state;
    a.ctrl = env->ctrl;
    a.out = env->out;
    a.ws = env->ws;
    a.contact_ids = env->contact_ids;
    a.n = env->n;
    a.p = env->p;
    static int lanes_cfg = 0;
    if (!lanes_cfg) {
        const char* s_env = getenv("V2P_LANES_PER_WAVE");
        lanes_cfg = s_env ? atoi(s_env) : 64;
        if (lanes_cfg < 1 || lanes_cfg > 64) lanes_cfg = 64;
    }
    a.lanes = lanes_cfg;
    unsigned blocks = (unsigned)((env->n + a.lanes - 1) / a.lanes);
    if (env->p.enable_contact)
        hipLaunchKernelGGL(physics_kernel<true>, dim3(blocks), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL(physics_kernel<false>, dim3(blocks), dim3(64), 0, s, a);
    return check_hip(hipGetLastError(), "physics_kernel");
}

}  // namespace v2p
