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
// Mapping: ONE ENVIRONMENT PER LANE, L environments per workgroup (one partly filled wave).
// The step is a long chain of dependent small-matrix operations per env, so what bounds it at
// 8192 envs is the latency of ONE wave, not throughput: the whole per-link working set (52
// floats x 24 links = 4992 B per env) therefore lives in LDS ([link][slot][lane], conflict
// free), L = 8 puts one wave on every SIMD of the chip (4 x 39 KB of LDS per CU), and global
// memory is touched only to stage the state in/out, for the per-body contact records and for
// Lambda_b, which are software-prefetched one body ahead.  The body model is identical for
// every lane and comes in through scalar loads; loops over links / vertices / rows are
// wave-uniform (scalar control flow, no divergence except value selects).
#include <math.h>
#include <stdlib.h>

#include "v2p_internal.hpp"
#include "v2p_dev.hpp"
#include "phys_common.hpp"

namespace v2p {

int physics_ws_slots() { return WS_SLOTS; }

// ---------------------------------------------------------------------------- the kernel
#define G(b, k) ws[((b) * LINK_SLOTS + (k)) * N + e]

// LDS record of one link: 13 float4 granules = 52 floats, [link][granule][lane] so that every access is a
// conflict-free ds_read/write_b128.  Lifetimes overlap by design (see the per-phase comments):
//   g0  Q.xyzw                         world quaternion
//   g1  X.xyz | I20                    origin position | last entry of the articulated inertia
//   g2  W.xyz | V.x                    link velocity (old -> v* -> solved)
//   g3  V.y V.z | P0 P1                P = articulated bias force -> acceleration -> delta-velocity of the impulse sweep
//   g4  P2..P5
//   g5  R.xyz | T.x                    R = x_b - x_parent ; T = joint torque (world axes)
//   g6  T.y T.z Z0 Z1                  Z = velocity-product terms;  after pass 3: g6 = (dw.xyz, dv.x)
//   g7  Z2..Z5                                                      g7 = (dv.y, dv.z) ; in the sweep g7.xyz = delta-u
//   g8..g12  I0..I19                   articulated inertia A(6) B(9) C(5 of 6) -> D^-1(6) E(9) u(3);
//                                      between substeps g8 = joint quaternion, g9.xyz = joint rate, g11.xyz = PD target
constexpr int NGRAN = 13;

struct LinkRec {  // what the impulse propagation needs of one link
    Sym3 Di;
    M3 E;
    V3 r;
    float aug;
};

struct BodyContacts {  // the per-body record the Gauss-Seidel block update needs (lives in global memory)
    Blocks L;
    V3 r[4];
    float bias[4];
    V3 lam[4];  // (n, t1, t2)
    int cnt;
};

__device__ __forceinline__ BodyContacts load_body(const float* __restrict__ ws, int64_t N, int64_t e, int b) {
    BodyContacts c;
    c.L.A = Sym3{G(b, GL), G(b, GL + 1), G(b, GL + 2), G(b, GL + 3), G(b, GL + 4), G(b, GL + 5)};
#pragma unroll
    for (int i = 0; i < 9; ++i) c.L.B.m[i] = G(b, GL + 6 + i);
    c.L.C = Sym3{G(b, GL + 15), G(b, GL + 16), G(b, GL + 17), G(b, GL + 18), G(b, GL + 19), G(b, GL + 20)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c.r[k] = V3{G(b, GCR + 3 * k), G(b, GCR + 3 * k + 1), G(b, GCR + 3 * k + 2)};
        c.bias[k] = G(b, GCB + k);
        c.lam[k] = V3{G(b, GCL + 3 * k), G(b, GCL + 3 * k + 1), G(b, GCL + 3 * k + 2)};
    }
    c.cnt = (int)G(b, GCN);
    return c;
}

__device__ __forceinline__ int unpack5(const unsigned long long* p, int i) {
    unsigned long long w = i < 12 ? p[0] : p[1];
    int k = i < 12 ? i : i - 12;
    return (int)((w >> (5 * k)) & 31ull);
}
__device__ __forceinline__ float4 f4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
__device__ __forceinline__ V3 xyz(float4 v) { return V3{v.x, v.y, v.z}; }

#define SUBPH(k) do { if (a.prof) { long long t_ = clock64(); if (blockIdx.x == 0 && lane == 0) atomicAdd((unsigned long long*)&a.prof[k], (unsigned long long)(t_ - tsub)); tsub = t_; } } while (0)
#define PHASE(k) do { if (a.prof) { long long t_ = clock64(); if (blockIdx.x == 0 && lane == 0) atomicAdd((unsigned long long*)&a.prof[k], (unsigned long long)(t_ - tprev)); tprev = t_; } } while (0)
#define SG(b, g) S[b][g][lane]

template <bool CONTACT, int L>
__global__ __launch_bounds__(L) void physics_kernel(PhysArgs a) {
    __shared__ float4 S[NB][NGRAN][L];
    const int64_t N = a.n;
    const int lane = threadIdx.x;
    int64_t e = (int64_t)blockIdx.x * L + lane;
    if (e >= N) e = N - 1;  // tail lanes shadow the last env (identical values, identical addresses)
    ConstModel& M = *(ConstModel*)a.model;
    float* __restrict__ ws = a.ws;
    float* __restrict__ st = a.state;
    const EnvParams& P = a.p;
    const float h = P.h;
    long long tprev = a.prof ? clock64() : 0;

    // D^-1, E, r of link i (5 + 1 wide LDS loads)
    auto load_rec = [&](int i) {
        float4 i0 = SG(i, 8), i1 = SG(i, 9), i2 = SG(i, 10), i3 = SG(i, 11), r = SG(i, 5);
        LinkRec o;
        o.Di = Sym3{i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
        o.E.m[0] = i1.z; o.E.m[1] = i1.w; o.E.m[2] = i2.x; o.E.m[3] = i2.y; o.E.m[4] = i2.z; o.E.m[5] = i2.w;
        o.E.m[6] = i3.x; o.E.m[7] = i3.y; o.E.m[8] = i3.z;
        o.r = xyz(r);
        o.aug = r.w;  // valid after pass 2
        return o;
    };
    auto ld_w = [&](int i) { return xyz(SG(i, 2)); };
    auto ld_q = [&](int i) { float4 q = SG(i, 0); return Q4{q.x, q.y, q.z, q.w}; };

    // ---- stage the generalized state into LDS (global state is read once and written once per launch)
    SG(0, 0) = f4(st[SIDX(ST_ROOT_QUAT + 0)], st[SIDX(ST_ROOT_QUAT + 1)], st[SIDX(ST_ROOT_QUAT + 2)], st[SIDX(ST_ROOT_QUAT + 3)]);
    SG(0, 1) = f4(st[SIDX(ST_ROOT_POS + 0)], st[SIDX(ST_ROOT_POS + 1)], st[SIDX(ST_ROOT_POS + 2)], 0.f);
    SG(0, 2) = f4(st[SIDX(ST_VEL + 3)], st[SIDX(ST_VEL + 4)], st[SIDX(ST_VEL + 5)], st[SIDX(ST_VEL + 0)]);
    SG(0, 3) = f4(st[SIDX(ST_VEL + 1)], st[SIDX(ST_VEL + 2)], 0.f, 0.f);
    for (int b = 1; b < NB; ++b) {
        const int jb = ST_JQUAT + 4 * (b - 1), vb = ST_VEL + 6 + 3 * (b - 1);
        SG(b, 8) = f4(st[SIDX(jb + 0)], st[SIDX(jb + 1)], st[SIDX(jb + 2)], st[SIDX(jb + 3)]);
        SG(b, 9) = f4(st[SIDX(vb + 0)], st[SIDX(vb + 1)], st[SIDX(vb + 2)], 0.f);
    }

    for (int sub = 0; sub < P.nsub; ++sub) {
        const bool wrench_on = sub < P.hold_sub;
        const bool last = sub == P.nsub - 1;
        for (int b = 1; b < NB; ++b) {
            const int cb = CT_PD + 3 * (b - 1);
            SG(b, 11) = f4(a.ctrl[CIDX(cb + 0)], a.ctrl[CIDX(cb + 1)], a.ctrl[CIDX(cb + 2)], 0.f);
        }
        PHASE(0);
        // ================================================================ pass 1: root -> leaves
        for (int b = 0; b < NB; ++b) {
            const int par = M.parents[b];
            Q4 q;
            V3 x, w, xd, zw{0.f, 0.f, 0.f}, zv{0.f, 0.f, 0.f}, tau{0.f, 0.f, 0.f}, r{0.f, 0.f, 0.f};
            if (b == 0) {
                float4 g2 = SG(0, 2), g3 = SG(0, 3);
                q = ld_q(0);
                x = xyz(SG(0, 1));
                w = xyz(g2);
                xd = V3{g2.w, g3.x, g3.y};
            } else {
                float4 jq4 = SG(b, 8), wt4 = SG(b, 9), tar4 = SG(b, 11);
                float4 pg2 = SG(par, 2), pg3 = SG(par, 3);
                Q4 jq{jq4.x, jq4.y, jq4.z, jq4.w};
                V3 wt = xyz(wt4), tar = xyz(tar4);
                Q4 qp = ld_q(par);
                V3 xp = xyz(SG(par, 1)), wp = xyz(pg2), xdp{pg2.w, pg3.x, pg3.y};
                q = qnormalize(qmul(qp, jq));
                M3 Rp = q2mat(qp);
                r = mul(Rp, V3{M.shape.local_pos[b][0], M.shape.local_pos[b][1], M.shape.local_pos[b][2]});
                x = xp + r;
                M3 Rb = q2mat(q);
                V3 wrel = mul(Rb, wt);
                w = wp + wrel;
                V3 wpr = cross(wp, r);
                xd = xdp + wpr;
                zw = cross(wp, wrel);
                zv = cross(wp, wpr);
                // implicit PD drive: kp (q_tar - q) - (kd + h kp) wrel, q = exp-map of the joint quaternion
                V3 qe = quat_to_expmap_stable(jq);
                float kp = M.shape.kp[b], kdh = M.shape.kd[b] + h * M.shape.kp[b];
                tau = mul(Rb, kp * (tar - qe) - kdh * wt);
            }
            M3 R = q2mat(q);
            // body inertia at its origin, world axes
            const float m = M.shape.mass[b];
            V3 d = mul(R, V3{M.shape.com[b][0], M.shape.com[b][1], M.shape.com[b][2]});
            Sym3 Ib{M.shape.inertia[b][0], M.shape.inertia[b][1], M.shape.inertia[b][2], M.shape.inertia[b][3], M.shape.inertia[b][4], M.shape.inertia[b][5]};
            V3 c0 = mul(Ib, V3{R.m[0], R.m[1], R.m[2]});  // Ic = R Ib R^T
            V3 c1 = mul(Ib, V3{R.m[3], R.m[4], R.m[5]});
            V3 c2 = mul(Ib, V3{R.m[6], R.m[7], R.m[8]});
            V3 r0 = row(R, 0), r1 = row(R, 1), r2 = row(R, 2);
            Sym3 Ic{dot(r0, c0), dot(r0, c1), dot(r0, c2), dot(r1, c1), dot(r1, c2), dot(r2, c2)};
            float dd = dot(d, d);
            Sym3 A{Ic.xx + m * (dd - d.x * d.x), Ic.xy - m * d.x * d.y, Ic.xz - m * d.x * d.z,
                   Ic.yy + m * (dd - d.y * d.y), Ic.yz - m * d.y * d.z, Ic.zz + m * (dd - d.z * d.z)};
            // bias force (velocity terms - gravity - external wrench)
            V3 wwd = cross(w, cross(w, d));
            V3 fl = m * (wwd - V3{0.f, 0.f, P.gravity_z});
            V3 nn = cross(w, mul(Ic, w)) + cross(d, fl);
            if (b == 0 && wrench_on) {
                V3 F{a.ctrl[CIDX(CT_FORCE + 0)], a.ctrl[CIDX(CT_FORCE + 1)], a.ctrl[CIDX(CT_FORCE + 2)]};
                V3 T{a.ctrl[CIDX(CT_TORQUE + 0)], a.ctrl[CIDX(CT_TORQUE + 1)], a.ctrl[CIDX(CT_TORQUE + 2)]};
                nn = nn - T - cross(d, F);  // force acts at the root COM
                fl = fl - F;
            }
            SG(b, 0) = f4(q.x, q.y, q.z, q.w);
            SG(b, 1) = f4(x.x, x.y, x.z, m);               // I20 = C.zz = m
            SG(b, 2) = f4(w.x, w.y, w.z, xd.x);
            SG(b, 3) = f4(xd.y, xd.z, nn.x, nn.y);
            SG(b, 4) = f4(nn.z, fl.x, fl.y, fl.z);
            SG(b, 5) = f4(r.x, r.y, r.z, tau.x);
            SG(b, 6) = f4(tau.y, tau.z, zw.x, zw.y);
            SG(b, 7) = f4(zw.z, zv.x, zv.y, zv.z);
            // IA: A(6) | B = m [d]x (9, row-major) | C = m 1 (6: xx xy xz yy yz zz)
            SG(b, 8) = f4(A.xx, A.xy, A.xz, A.yy);
            SG(b, 9) = f4(A.yz, A.zz, 0.f, -m * d.z);
            SG(b, 10) = f4(m * d.y, m * d.z, 0.f, -m * d.x);
            SG(b, 11) = f4(-m * d.y, m * d.x, 0.f, m);
            SG(b, 12) = f4(0.f, 0.f, m, 0.f);
        }

        PHASE(1);
        // ================================================================ pass 2: leaves -> root
        for (int b = NB - 1; b >= 1; --b) {
            const int par = M.parents[b];
            const float aug = P.aug[b];
            float4 i0 = SG(b, 8), i1 = SG(b, 9), i2 = SG(b, 10), i3 = SG(b, 11), i4 = SG(b, 12), g1 = SG(b, 1);
            float4 g3 = SG(b, 3), g4 = SG(b, 4), g5 = SG(b, 5), g6 = SG(b, 6), g7 = SG(b, 7);
            // parent accumulators, loaded early (independent of the math below)
            float4 p0 = SG(par, 8), p1 = SG(par, 9), p2 = SG(par, 10), p3 = SG(par, 11), p4 = SG(par, 12), pg1 = SG(par, 1), pg3 = SG(par, 3), pg4 = SG(par, 4);
            Sym3 A{i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
            M3 B;
            B.m[0] = i1.z; B.m[1] = i1.w; B.m[2] = i2.x; B.m[3] = i2.y; B.m[4] = i2.z; B.m[5] = i2.w; B.m[6] = i3.x; B.m[7] = i3.y; B.m[8] = i3.z;
            Sym3 C{i3.w, i4.x, i4.y, i4.z, i4.w, g1.w};
            V3 pn{g3.z, g3.w, g4.x}, pf{g4.y, g4.z, g4.w};
            V3 r = xyz(g5), tau{g5.w, g6.x, g6.y}, zw{g6.z, g6.w, g7.x}, zv{g7.y, g7.z, g7.w};
            Sym3 D{A.xx + aug, A.xy, A.xz, A.yy + aug, A.yz, A.zz + aug};
            Sym3 Di = inv(D);
            M3 E = mul(Di, B);
            V3 u = tau - pn;
            // the articulated inertia of b is dead from here on: its slots now hold D^-1, E, u
            SG(b, 8) = f4(Di.xx, Di.xy, Di.xz, Di.yy);
            SG(b, 9) = f4(Di.yz, Di.zz, E.m[0], E.m[1]);
            SG(b, 10) = f4(E.m[2], E.m[3], E.m[4], E.m[5]);
            SG(b, 11) = f4(E.m[6], E.m[7], E.m[8], u.x);
            SG(b, 12) = f4(u.y, u.z, 0.f, 0.f);
            SG(b, 5) = f4(r.x, r.y, r.z, aug);  // the joint torque is consumed: keep the diagonal augmentation next to r
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
            // shift to the parent origin and accumulate: S = [r]x Ca (columns r x Ca_col)
            V3 s0 = cross(r, V3{Ca.xx, Ca.xy, Ca.xz}), s1 = cross(r, V3{Ca.xy, Ca.yy, Ca.yz}), s2 = cross(r, V3{Ca.xz, Ca.yz, Ca.zz});
            M3 Y;  // parent B increment = Ba + S
            Y.m[0] = Ba.m[0] + s0.x; Y.m[1] = Ba.m[1] + s1.x; Y.m[2] = Ba.m[2] + s2.x;
            Y.m[3] = Ba.m[3] + s0.y; Y.m[4] = Ba.m[4] + s1.y; Y.m[5] = Ba.m[5] + s2.y;
            Y.m[6] = Ba.m[6] + s0.z; Y.m[7] = Ba.m[7] + s1.z; Y.m[8] = Ba.m[8] + s2.z;
            V3 t10 = cross(r, row(Ba, 0)), t11 = cross(r, row(Ba, 1)), t12 = cross(r, row(Ba, 2));  // T1 rows = r x Ba_row
            V3 sr0{s0.x, s1.x, s2.x}, sr1{s0.y, s1.y, s2.y}, sr2{s0.z, s1.z, s2.z};
            V3 t20 = cross(r, sr0), t21 = cross(r, sr1), t22 = cross(r, sr2);                        // T2 rows = r x S_row
            V3 pinc_n = pan + cross(r, paf);
            p0.x += Aa.xx + 2.f * t10.x + t20.x;
            p0.y += Aa.xy + t10.y + t11.x + t20.y;
            p0.z += Aa.xz + t10.z + t12.x + t20.z;
            p0.w += Aa.yy + 2.f * t11.y + t21.y;
            p1.x += Aa.yz + t11.z + t12.y + t21.z;
            p1.y += Aa.zz + 2.f * t12.z + t22.z;
            p1.z += Y.m[0]; p1.w += Y.m[1]; p2.x += Y.m[2]; p2.y += Y.m[3]; p2.z += Y.m[4]; p2.w += Y.m[5]; p3.x += Y.m[6]; p3.y += Y.m[7]; p3.z += Y.m[8];
            p3.w += Ca.xx; p4.x += Ca.xy; p4.y += Ca.xz; p4.z += Ca.yy; p4.w += Ca.yz; pg1.w += Ca.zz;
            pg3.z += pinc_n.x; pg3.w += pinc_n.y; pg4.x += pinc_n.z; pg4.y += paf.x; pg4.z += paf.y; pg4.w += paf.z;
            SG(par, 8) = p0; SG(par, 9) = p1; SG(par, 10) = p2; SG(par, 11) = p3; SG(par, 12) = p4; SG(par, 1) = pg1; SG(par, 3) = pg3; SG(par, 4) = pg4;
        }

        PHASE(2);
        // ================================================================ root: 6x6 solve
        Blocks Lsave[MAX_BRANCH];  // Lambda of the branching links (root, chest): children that do not follow their parent read it here
        {
            float4 i0 = SG(0, 8), i1 = SG(0, 9), i2 = SG(0, 10), i3 = SG(0, 11), i4 = SG(0, 12), g1 = SG(0, 1), g2 = SG(0, 2), g3 = SG(0, 3), g4 = SG(0, 4);
            Blocks I0;
            I0.A = Sym3{i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
            I0.B.m[0] = i1.z; I0.B.m[1] = i1.w; I0.B.m[2] = i2.x; I0.B.m[3] = i2.y; I0.B.m[4] = i2.z; I0.B.m[5] = i2.w; I0.B.m[6] = i3.x; I0.B.m[7] = i3.y; I0.B.m[8] = i3.z;
            I0.C = Sym3{i3.w, i4.x, i4.y, i4.z, i4.w, g1.w};
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
#pragma unroll
            for (int s = 0; s < MAX_BRANCH; ++s) Lsave[s] = L0;
            V3 pn{g3.z, g3.w, g4.x}, pf{g4.y, g4.z, g4.w};
            // a0 = -Lambda_0 p
            V3 al = -(mul(L0.A, pn) + mul(L0.B, pf));
            V3 ac = -(V3{dot(col(L0.B, 0), pn), dot(col(L0.B, 1), pn), dot(col(L0.B, 2), pn)} + mul(L0.C, pf));
            V3 w = xyz(g2) + h * al;
            V3 xd = V3{g2.w, g3.x, g3.y} + h * ac;
            SG(0, 2) = f4(w.x, w.y, w.z, xd.x);
            SG(0, 3) = f4(xd.y, xd.z, al.x, al.y);
            SG(0, 4) = f4(al.z, ac.x, ac.y, ac.z);
            SG(0, 6) = f4(h * al.x, h * al.y, h * al.z, h * ac.x);  // v* - v_old of the root
            SG(0, 7) = f4(h * ac.y, h * ac.z, 0.f, 0.f);
        }

        // ================================================================ pass 3: root -> leaves (accelerations, v*)
        for (int b = 1; b < NB; ++b) {
            const int par = M.parents[b];
            const float aug = P.aug[b];
            float4 pg3 = SG(par, 3), pg4 = SG(par, 4), pg6 = SG(par, 6), pg7 = SG(par, 7);
            float4 g2 = SG(b, 2), g3 = SG(b, 3), g5 = SG(b, 5), g6 = SG(b, 6), g7 = SG(b, 7);
            float4 i0 = SG(b, 8), i1 = SG(b, 9), i2 = SG(b, 10), i3 = SG(b, 11), i4 = SG(b, 12);
            V3 alp{pg3.z, pg3.w, pg4.x}, acp{pg4.y, pg4.z, pg4.w};
            V3 dwp = xyz(pg6), dvp{pg6.w, pg7.x, pg7.y};  // parent's (v* - v_old)
            V3 r = xyz(g5), zw{g6.z, g6.w, g7.x}, zv{g7.y, g7.z, g7.w};
            Sym3 Di{i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
            M3 E;
            E.m[0] = i1.z; E.m[1] = i1.w; E.m[2] = i2.x; E.m[3] = i2.y; E.m[4] = i2.z; E.m[5] = i2.w; E.m[6] = i3.x; E.m[7] = i3.y; E.m[8] = i3.z;
            V3 u{i3.w, i4.x, i4.y};
            V3 aw = alp + zw;
            V3 av = acp + cross(alp, r) + zv;
            V3 qdd = mul(Di, u + aug * aw) - aw - mul(E, av);
            V3 al = aw + qdd;
            // unconstrained velocity at the OLD configuration, as an increment: dw_b = dw_p + h qdd, dxd_b = dxd_p + dw_p x r
            V3 dw = dwp + h * qdd;
            V3 dv = dvp + cross(dwp, r);
            V3 w = xyz(g2) + dw;
            V3 xd = V3{g2.w, g3.x, g3.y} + dv;
            SG(b, 2) = f4(w.x, w.y, w.z, xd.x);
            SG(b, 3) = f4(xd.y, xd.z, al.x, al.y);
            SG(b, 4) = f4(al.z, av.x, av.y, av.z);
            SG(b, 6) = f4(dw.x, dw.y, dw.z, dv.x);
            SG(b, 7) = f4(dv.y, dv.z, 0.f, 0.f);
        }

        PHASE(3);
        if (CONTACT) {
            // ============================================================ contact generation
            const float coff = P.contact_offset;
            unsigned touch = 0;  // bodies that have a contact in at least one env of this wave (wave-uniform)
            for (int b = 0; b < NB; ++b) {
                V3 x = xyz(SG(b, 1));
                int cnt = 0;
                bool near = x.z - M.shape.bound_radius[b] < coff;
                if (__any(near)) {
                    M3 R = q2mat(ld_q(b));
                    const int v0 = M.shape.hull_offsets[b], nv = M.shape.hull_count[b];
                    int f0 = -1, f1 = -1, f2 = -1, f3 = -1, k0 = -1;
                    float zmin = 0.f;
                    // vertex lists are padded to a multiple of HULL_PAD: 8 vertices = 24 consecutive floats per batch of scalar loads
                    for (int vb = 0; vb < nv; vb += HULL_PAD) {
                        float hv[3 * HULL_PAD];
#pragma unroll
                        for (int k = 0; k < 3 * HULL_PAD; ++k) hv[k] = (&M.shape.hull_verts[v0 + vb][0])[k];
#pragma unroll
                        for (int j = 0; j < HULL_PAD; ++j) {
                            const int i = vb + j;
                            float z = x.z + R.m[6] * hv[3 * j] + R.m[7] * hv[3 * j + 1] + R.m[8] * hv[3 * j + 2];
                            bool c = (z < coff) && (i < nv);
                            f0 = (c && cnt == 0) ? i : f0;
                            f1 = (c && cnt == 1) ? i : f1;
                            f2 = (c && cnt == 2) ? i : f2;
                            f3 = (c && cnt == 3) ? i : f3;
                            bool better = c && (k0 < 0 || z < zmin);
                            k0 = better ? i : k0;
                            zmin = better ? z : zmin;
                            cnt += c ? 1 : 0;
                        }
                    }
                    int s0 = f0, s1 = f1, s2 = f2, s3 = f3, ns = cnt < 4 ? cnt : 4;
                    if (__any(cnt > 4)) {
                        // manifold reduction: deepest, farthest from it, extreme on either side of that line
                        int kk0 = k0 < 0 ? 0 : k0;
                        V3 u0{M.shape.hull_verts[v0 + kk0][0], M.shape.hull_verts[v0 + kk0][1], M.shape.hull_verts[v0 + kk0][2]};
                        float p0x = x.x + R.m[0] * u0.x + R.m[1] * u0.y + R.m[2] * u0.z;
                        float p0y = x.y + R.m[3] * u0.x + R.m[4] * u0.y + R.m[5] * u0.z;
                        int k1 = -1;
                        float best = -1.f;
                        for (int vb = 0; vb < nv; vb += HULL_PAD) {
                            float hv[3 * HULL_PAD];
#pragma unroll
                            for (int k = 0; k < 3 * HULL_PAD; ++k) hv[k] = (&M.shape.hull_verts[v0 + vb][0])[k];
#pragma unroll
                            for (int j = 0; j < HULL_PAD; ++j) {
                                const int i = vb + j;
                                float ux = hv[3 * j], uy = hv[3 * j + 1], uz = hv[3 * j + 2];
                                float z = x.z + R.m[6] * ux + R.m[7] * uy + R.m[8] * uz;
                                float dx = x.x + R.m[0] * ux + R.m[1] * uy + R.m[2] * uz - p0x;
                                float dy = x.y + R.m[3] * ux + R.m[4] * uy + R.m[5] * uz - p0y;
                                float d2 = dx * dx + dy * dy;
                                bool take = (z < coff) && (i < nv) && (i != k0) && (d2 > best);
                                best = take ? d2 : best;
                                k1 = take ? i : k1;
                            }
                        }
                        int kk1 = k1 < 0 ? 0 : k1;
                        V3 u1{M.shape.hull_verts[v0 + kk1][0], M.shape.hull_verts[v0 + kk1][1], M.shape.hull_verts[v0 + kk1][2]};
                        float ex = x.x + R.m[0] * u1.x + R.m[1] * u1.y + R.m[2] * u1.z - p0x;
                        float ey = x.y + R.m[3] * u1.x + R.m[4] * u1.y + R.m[5] * u1.z - p0y;
                        int k2 = -1, k3 = -1;
                        float amax = 0.f, amin = 0.f;
                        for (int vb = 0; vb < nv; vb += HULL_PAD) {
                            float hv[3 * HULL_PAD];
#pragma unroll
                            for (int k = 0; k < 3 * HULL_PAD; ++k) hv[k] = (&M.shape.hull_verts[v0 + vb][0])[k];
#pragma unroll
                            for (int j = 0; j < HULL_PAD; ++j) {
                                const int i = vb + j;
                                float ux = hv[3 * j], uy = hv[3 * j + 1], uz = hv[3 * j + 2];
                                float z = x.z + R.m[6] * ux + R.m[7] * uy + R.m[8] * uz;
                                float dx = x.x + R.m[0] * ux + R.m[1] * uy + R.m[2] * uz - p0x;
                                float dy = x.y + R.m[3] * ux + R.m[4] * uy + R.m[5] * uz - p0y;
                                float area = ex * dy - ey * dx;
                                bool cand = (z < coff) && (i < nv) && (i != k0) && (i != k1);
                                bool up = cand && area > amax;
                                bool dn = cand && area < amin;
                                amax = up ? area : amax; k2 = up ? i : k2;
                                amin = dn ? area : amin; k3 = dn ? i : k3;
                            }
                        }
                        if (cnt > 4) {
                            s0 = k0; s1 = k1;
                            s2 = k2 >= 0 ? k2 : k3;
                            s3 = k2 >= 0 ? k3 : -1;
                            ns = 2 + (k2 >= 0 ? 1 : 0) + (k3 >= 0 ? 1 : 0);
                        }
                    }
                    cnt = ns;
                    int sel[4] = {s0, s1, s2, s3};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        int vi = sel[c] < 0 ? 0 : sel[c];
                        V3 rr = mul(R, V3{M.shape.hull_verts[v0 + vi][0], M.shape.hull_verts[v0 + vi][1], M.shape.hull_verts[v0 + vi][2]});
                        G(b, GCR + 3 * c) = rr.x; G(b, GCR + 3 * c + 1) = rr.y; G(b, GCR + 3 * c + 2) = rr.z;
                        float d = x.z + rr.z - P.rest_offset;
                        G(b, GCB + c) = d >= 0.f ? d / h : fmaxf(P.erp * d / h, -P.max_depen);
                        G(b, GCL + 3 * c) = 0.f; G(b, GCL + 3 * c + 1) = 0.f; G(b, GCL + 3 * c + 2) = 0.f;
                        if (a.contact_ids) a.contact_ids[(e * NB + b) * 4 + c] = c < cnt ? b * 64 + sel[c] : -1;
                        if (a.contact_ids_sub) a.contact_ids_sub[((e * P.nsub + sub) * NB + b) * 4 + c] = c < cnt ? b * 64 + sel[c] : -1;
                    }
                    if (__any(cnt > 0)) touch |= 1u << b;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (a.contact_ids) a.contact_ids[(e * NB + b) * 4 + c] = -1;
                        if (a.contact_ids_sub) a.contact_ids_sub[((e * P.nsub + sub) * NB + b) * 4 + c] = -1;
                    }
                }
                G(b, GCN) = (float)cnt;
            }

            PHASE(4);
            // ============================================================ Lambda_b recursion (root -> leaves)
            {
                Blocks Lprev = Lsave[0];
                {  // Lambda_0 to global
                    G(0, GL + 0) = Lprev.A.xx; G(0, GL + 1) = Lprev.A.xy; G(0, GL + 2) = Lprev.A.xz; G(0, GL + 3) = Lprev.A.yy; G(0, GL + 4) = Lprev.A.yz; G(0, GL + 5) = Lprev.A.zz;
#pragma unroll
                    for (int i = 0; i < 9; ++i) G(0, GL + 6 + i) = Lprev.B.m[i];
                    G(0, GL + 15) = Lprev.C.xx; G(0, GL + 16) = Lprev.C.xy; G(0, GL + 17) = Lprev.C.xz; G(0, GL + 18) = Lprev.C.yy; G(0, GL + 19) = Lprev.C.yz; G(0, GL + 20) = Lprev.C.zz;
                }
                unsigned anc = 0;  // links whose Lambda is required = ancestors-or-self of touched bodies
                for (int b = NB - 1; b >= 1; --b)
                    if (((touch | anc) >> b) & 1u) anc |= (1u << b) | (1u << M.parents[b]);
                for (int b = 1; b < NB; ++b) {
                    const int par = M.parents[b];
                    const int pslot = M.lam_slot[par];
                    if (!((anc >> b) & 1u)) continue;  // wave-uniform: nobody in this wave touches b or anything below it
                    // parent's Lambda: the previous iteration for a chain link, else one of the saved sets (constant register indices)
                    Blocks Lp = Lprev;
                    if (par != b - 1) Lp = pslot == 1 ? Lsave[1] : (pslot == 2 ? Lsave[2] : Lsave[0]);
                    const float aug = P.aug[b];
                    LinkRec rec = load_rec(b);
                    const V3 r = rec.r;
                    const Sym3 Di = rec.Di;
                    const M3 E = rec.E;
                    // G = X Lp X^T: Ga = La ; Gb = La [r]x + Lb ; Gc = Lc - [r]x Lb + Gb^T [r]x
                    V3 la0{Lp.A.xx, Lp.A.xy, Lp.A.xz}, la1{Lp.A.xy, Lp.A.yy, Lp.A.yz}, la2{Lp.A.xz, Lp.A.yz, Lp.A.zz};
                    M3 Gb;
                    {
                        V3 g0 = cross(la0, r) + row(Lp.B, 0), g1 = cross(la1, r) + row(Lp.B, 1), g2 = cross(la2, r) + row(Lp.B, 2);
                        Gb.m[0] = g0.x; Gb.m[1] = g0.y; Gb.m[2] = g0.z; Gb.m[3] = g1.x; Gb.m[4] = g1.y; Gb.m[5] = g1.z; Gb.m[6] = g2.x; Gb.m[7] = g2.y; Gb.m[8] = g2.z;
                    }
                    Sym3 Gc;
                    {
                        V3 q0 = cross(col(Gb, 0), r), q1 = cross(col(Gb, 1), r), q2 = cross(col(Gb, 2), r);        // rows of Gb^T [r]x
                        V3 m0 = cross(r, col(Lp.B, 0)), m1 = cross(r, col(Lp.B, 1)), m2 = cross(r, col(Lp.B, 2));  // columns of [r]x Lb
                        Gc.xx = Lp.C.xx + q0.x - m0.x;
                        Gc.xy = Lp.C.xy + q0.y - m1.x;
                        Gc.xz = Lp.C.xz + q0.z - m2.x;
                        Gc.yy = Lp.C.yy + q1.y - m1.y;
                        Gc.yz = Lp.C.yz + q1.z - m2.y;
                        Gc.zz = Lp.C.zz + q2.z - m2.z;
                    }
                    // Lambda_b = [Di 0; 0 0] + T^T G T,  T = [aug Di, 0; -E^T, 1]
                    M3 DiM;
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
                            V3 ej = row(E, j);
                            H1.m[3 * i + j] = aug * dot(gai, dj) - dot(gbi, ej);
                            H2.m[3 * i + j] = aug * dot(gbti, dj) - dot(gci, ej);
                        }
                    }
                    Blocks Lb;
                    {
                        M3 t;
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 3; ++j) t.m[3 * i + j] = aug * dot(row(DiM, i), col(H1, j)) - dot(row(E, i), col(H2, j));
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
                    Lprev = Lb;
                    const int slot = M.lam_slot[b];
#pragma unroll
                    for (int s = 1; s < MAX_BRANCH; ++s)
                        if (slot == s) Lsave[s] = Lb;
                    if ((touch >> b) & 1u) {
                        G(b, GL + 0) = Lb.A.xx; G(b, GL + 1) = Lb.A.xy; G(b, GL + 2) = Lb.A.xz; G(b, GL + 3) = Lb.A.yy; G(b, GL + 4) = Lb.A.yz; G(b, GL + 5) = Lb.A.zz;
#pragma unroll
                        for (int i = 0; i < 9; ++i) G(b, GL + 6 + i) = Lb.B.m[i];
                        G(b, GL + 15) = Lb.C.xx; G(b, GL + 16) = Lb.C.xy; G(b, GL + 17) = Lb.C.xz; G(b, GL + 18) = Lb.C.yy; G(b, GL + 19) = Lb.C.yz; G(b, GL + 20) = Lb.C.zz;
                    }
                }
            }

            PHASE(5);
            // ============================================================ block Gauss-Seidel over the touched bodies
            if (a.prof && blockIdx.x == 0 && lane == 0) { atomicAdd((unsigned long long*)&a.prof[9], (unsigned long long)__popc(touch)); atomicAdd((unsigned long long*)&a.prof[10], 1ull); }
            if (touch && P.n_iter > 0) {
                for (int it = 0; it < P.n_iter; ++it) {
                    unsigned todo = touch;
                    while (todo) {
                        const int b = __ffs(todo) - 1;
                        todo &= todo - 1;
                        if (a.prof && blockIdx.x == 0 && lane == 0) atomicAdd((unsigned long long*)&a.prof[8], 1ull);
                        const BodyContacts cur0 = load_body(ws, N, e, b);
                        BodyContacts cur = cur0;
                        long long tsub = a.prof ? clock64() : 0;
                        float4 bg2 = SG(b, 2), bg3 = SG(b, 3);
                        V3 w = xyz(bg2), xd{bg2.w, bg3.x, bg3.y};
                        LinkRec rec = load_rec(b);  // first link of the leaf -> root path: in flight while the rows are solved
                        V3 phin{0.f, 0.f, 0.f}, phif{0.f, 0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const bool active = c < cur.cnt;
                            V3 rr = cur.r[c];
                            float ln = cur.lam[c].x, l1 = cur.lam[c].y, l2 = cur.lam[c].z;
#pragma unroll
                            for (int ax = 0; ax < 3; ++ax) {
                                V3 dir = ax == 0 ? V3{0.f, 0.f, 1.f} : (ax == 1 ? V3{1.f, 0.f, 0.f} : V3{0.f, 1.f, 0.f});
                                V3 jn = cross(rr, dir);  // row = [jn ; dir] against (w, xd)
                                V3 yw = mul(cur.L.A, jn) + mul(cur.L.B, dir);
                                V3 yv = V3{dot(col(cur.L.B, 0), jn), dot(col(cur.L.B, 1), jn), dot(col(cur.L.B, 2), jn)} + mul(cur.L.C, dir);
                                float wii = dot(jn, yw) + dot(dir, yv);
                                float rel = dot(jn, w) + dot(dir, xd) + (ax == 0 ? cur.bias[c] : 0.f);
                                float old = ax == 0 ? ln : (ax == 1 ? l1 : l2);
                                float nl = old - rel * __builtin_amdgcn_rcpf(wii);
                                if (ax == 0) nl = fmaxf(nl, 0.f);
                                else { float lim = P.mu * ln; nl = fminf(fmaxf(nl, -lim), lim); }
                                float dl = active ? nl - old : 0.f;
                                if (ax == 0) ln += dl; else if (ax == 1) l1 += dl; else l2 += dl;
                                w = w + dl * yw;
                                xd = xd + dl * yv;
                                phin = phin + dl * jn;
                                phif = phif + dl * dir;
                            }
                            cur.lam[c] = V3{ln, l1, l2};
                            G(b, GCL + 3 * c) = ln; G(b, GCL + 3 * c + 1) = l1; G(b, GCL + 3 * c + 2) = l2;
                        }
                        SUBPH(11);
                        // ---- propagate the net impulse (phin, phif) applied at link b: leaf -> root
                        V3 nI = phin, fI = phif;
                        unsigned path = 0;
                        for (int i = b; i != 0;) {
                            const int par = unpack5(a.par_pack, i);
                            const float aug = rec.aug;
                            path |= 1u << i;
                            LinkRec prec = rec;
                            if (par != 0) prec = load_rec(par);  // prefetch the next link of the path
                            SG(i, 7) = f4(nI.x, nI.y, nI.z, 0.f);  // delta-u of the links on the path
                            V3 na = aug * mul(rec.Di, nI);
                            V3 fa = fI - V3{dot(col(rec.E, 0), nI), dot(col(rec.E, 1), nI), dot(col(rec.E, 2), nI)};
                            nI = na + cross(rec.r, fa);
                            fI = fa;
                            i = par;
                            rec = prec;
                        }
                        SUBPH(12);
                        // root response
                        {
                            const Blocks& L0 = Lsave[0];
                            float4 g2 = SG(0, 2), g3 = SG(0, 3);
                            V3 dw = mul(L0.A, nI) + mul(L0.B, fI);
                            V3 dv = V3{dot(col(L0.B, 0), nI), dot(col(L0.B, 1), nI), dot(col(L0.B, 2), nI)} + mul(L0.C, fI);
                            SG(0, 2) = f4(g2.x + dw.x, g2.y + dw.y, g2.z + dw.z, g2.w + dv.x);
                            SG(0, 3) = f4(g3.x + dv.y, g3.y + dv.z, dw.x, dw.y);
                            SG(0, 4) = f4(dw.z, dv.x, dv.y, dv.z);
                        }
                        SUBPH(13);
                        // root -> leaves in level order: every link moves; the next link's record is prefetched
                        {
                            int i = unpack5(a.ord_pack, 1);
                            LinkRec fr = load_rec(i);
                            float4 g2 = SG(i, 2), g3 = SG(i, 3);
                            for (int k = 1; k < NB; ++k) {
                                const int par = unpack5(a.par_pack, i);
                                const int inext = unpack5(a.ord_pack, k + 1 < NB ? k + 1 : k);
                                const float aug = fr.aug;
                                float4 pg3 = SG(par, 3), pg4 = SG(par, 4);
                                V3 dwp{pg3.z, pg3.w, pg4.x}, dvp{pg4.y, pg4.z, pg4.w};
                                V3 du{0.f, 0.f, 0.f};
                                if ((path >> i) & 1u) du = xyz(SG(i, 7));
                                LinkRec nr = load_rec(inext);
                                float4 ng2 = SG(inext, 2), ng3 = SG(inext, 3);
                                V3 av = dvp + cross(dwp, fr.r);
                                V3 dw = mul(fr.Di, aug * dwp + du) - mul(fr.E, av);
                                SG(i, 2) = f4(g2.x + dw.x, g2.y + dw.y, g2.z + dw.z, g2.w + av.x);
                                SG(i, 3) = f4(g3.x + av.y, g3.y + av.z, dw.x, dw.y);
                                SG(i, 4) = f4(dw.z, av.x, av.y, av.z);
                                i = inext; fr = nr; g2 = ng2; g3 = ng3;
                            }
                        }
                        SUBPH(14);
                    }
                }
            }
        }

        PHASE(6);
        // ================================================================ velocities -> generalized, damping, clamp, integrate
        const float sc = 1.f / (1.f + h * P.ang_damp);
        const float wmax = P.max_ang_vel;
        // joints first: they need the undamped link velocities of both ends
        for (int b = NB - 1; b >= 1; --b) {
            const int par = M.parents[b];
            Q4 qb = ld_q(b), qp = ld_q(par);
            M3 R = q2mat(qb);
            V3 wt = mulT(R, ld_w(b) - ld_w(par));     // joint rate, body axes (undamped)
            Q4 jq = qnormalize(qmul(qconj(qp), qb));  // joint quaternion of the old configuration
            if (last) {
                // joint drive torque actually applied over the substep (implicit form)
                V3 qe = quat_to_expmap_stable(jq);
                const int cb = CT_PD + 3 * (b - 1);
                V3 tar{a.ctrl[CIDX(cb + 0)], a.ctrl[CIDX(cb + 1)], a.ctrl[CIDX(cb + 2)]};
                V3 tf = M.shape.kp[b] * (tar - qe - h * wt) - M.shape.kd[b] * wt;
                const int ob = OUT_DOF_FORCE + 3 * (b - 1);
                a.out[OIDX(ob + 0)] = tf.x; a.out[OIDX(ob + 1)] = tf.y; a.out[OIDX(ob + 2)] = tf.z;
            }
            wt = sc * wt;
            float n2 = dot(wt, wt);
            if (n2 > wmax * wmax) wt = (wmax * rsqrtf(n2)) * wt;
            Q4 nq = qnormalize(qmul(jq, rotvec_to_quat(h * wt)));  // body-frame rate: right multiply
            SG(b, 8) = f4(nq.x, nq.y, nq.z, nq.w);
            SG(b, 9) = f4(wt.x, wt.y, wt.z, 0.f);
        }
        {
            float4 g1 = SG(0, 1), g2 = SG(0, 2), g3 = SG(0, 3);
            V3 w0 = sc * xyz(g2);
            V3 xd0{g2.w, g3.x, g3.y};
            float n2 = dot(w0, w0);
            if (n2 > wmax * wmax) w0 = (wmax * rsqrtf(n2)) * w0;
            Q4 nq = qnormalize(qmul(rotvec_to_quat(h * w0), ld_q(0)));  // world-frame rate: left multiply
            SG(0, 2) = f4(w0.x, w0.y, w0.z, xd0.x);
            SG(0, 1) = f4(g1.x + h * xd0.x, g1.y + h * xd0.y, g1.z + h * xd0.z, 0.f);
            SG(0, 0) = f4(nq.x, nq.y, nq.z, nq.w);
        }
        if (CONTACT && last) {
            // net contact force per body = sum of impulses / h  (refresh_net_contact_force_tensor)
            const float ih = 1.f / h;
            for (int b = 0; b < NB; ++b) {
                int cnt = (int)G(b, GCN);
                V3 f{0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < cnt) { f.z += G(b, GCL + 3 * c); f.x += G(b, GCL + 3 * c + 1); f.y += G(b, GCL + 3 * c + 2); }
                a.out[OIDX(OUT_CONTACT + 3 * b + 0)] = f.x * ih;
                a.out[OIDX(OUT_CONTACT + 3 * b + 1)] = f.y * ih;
                a.out[OIDX(OUT_CONTACT + 3 * b + 2)] = f.z * ih;
            }
        }
    }

    PHASE(7);
    // ==================================================================== final kinematics -> state, rigid-body state, dof_pos
    for (int b = 0; b < NB; ++b) {
        const int par = M.parents[b];
        Q4 q;
        V3 x, w, xd;
        if (b == 0) {
            float4 g2 = SG(0, 2), g3 = SG(0, 3);
            q = ld_q(0);
            x = xyz(SG(0, 1));
            w = xyz(g2);
            xd = V3{g2.w, g3.x, g3.y};
            st[SIDX(ST_ROOT_QUAT + 0)] = q.x; st[SIDX(ST_ROOT_QUAT + 1)] = q.y; st[SIDX(ST_ROOT_QUAT + 2)] = q.z; st[SIDX(ST_ROOT_QUAT + 3)] = q.w;
            st[SIDX(ST_ROOT_POS + 0)] = x.x; st[SIDX(ST_ROOT_POS + 1)] = x.y; st[SIDX(ST_ROOT_POS + 2)] = x.z;
            st[SIDX(ST_VEL + 0)] = xd.x; st[SIDX(ST_VEL + 1)] = xd.y; st[SIDX(ST_VEL + 2)] = xd.z;
            st[SIDX(ST_VEL + 3)] = w.x; st[SIDX(ST_VEL + 4)] = w.y; st[SIDX(ST_VEL + 5)] = w.z;
        } else {
            float4 jq4 = SG(b, 8), wt4 = SG(b, 9), pg2 = SG(par, 2), pg3 = SG(par, 3);
            Q4 jq{jq4.x, jq4.y, jq4.z, jq4.w};
            V3 wt = xyz(wt4);
            const int jb = ST_JQUAT + 4 * (b - 1), vb = ST_VEL + 6 + 3 * (b - 1);
            st[SIDX(jb + 0)] = jq.x; st[SIDX(jb + 1)] = jq.y; st[SIDX(jb + 2)] = jq.z; st[SIDX(jb + 3)] = jq.w;
            st[SIDX(vb + 0)] = wt.x; st[SIDX(vb + 1)] = wt.y; st[SIDX(vb + 2)] = wt.z;
            Q4 qp = ld_q(par);
            V3 xp = xyz(SG(par, 1)), wp = xyz(pg2), xdp{pg2.w, pg3.x, pg3.y};
            q = qnormalize(qmul(qp, jq));
            V3 r = mul(q2mat(qp), V3{M.shape.local_pos[b][0], M.shape.local_pos[b][1], M.shape.local_pos[b][2]});
            x = xp + r;
            w = wp + mul(q2mat(q), wt);
            xd = xdp + cross(wp, r);
            V3 qe = quat_to_expmap_stable(jq);
            const int ob = OUT_DOF_POS + 3 * (b - 1);
            a.out[OIDX(ob + 0)] = qe.x; a.out[OIDX(ob + 1)] = qe.y; a.out[OIDX(ob + 2)] = qe.z;
            SG(b, 0) = f4(q.x, q.y, q.z, q.w);
            SG(b, 1) = f4(x.x, x.y, x.z, 0.f);
            SG(b, 2) = f4(w.x, w.y, w.z, xd.x);
            SG(b, 3) = f4(xd.y, xd.z, 0.f, 0.f);
        }
        const int ob = OUT_RB + 13 * b;
        a.out[OIDX(ob + 0)] = x.x; a.out[OIDX(ob + 1)] = x.y; a.out[OIDX(ob + 2)] = x.z;
        a.out[OIDX(ob + 3)] = q.x; a.out[OIDX(ob + 4)] = q.y; a.out[OIDX(ob + 5)] = q.z; a.out[OIDX(ob + 6)] = q.w;
        a.out[OIDX(ob + 7)] = xd.x; a.out[OIDX(ob + 8)] = xd.y; a.out[OIDX(ob + 9)] = xd.z;
        a.out[OIDX(ob + 10)] = w.x; a.out[OIDX(ob + 11)] = w.y; a.out[OIDX(ob + 12)] = w.z;
    }
    if (!CONTACT) {
        for (int k = 0; k < NB * 3; ++k) a.out[OIDX(OUT_CONTACT + k)] = 0.f;
    }
}

template <int L>
static void launch_L(const PhysArgs& a, bool contact, hipStream_t s) {
    unsigned blocks = (unsigned)((a.n + L - 1) / L);
    if (contact)
        hipLaunchKernelGGL((physics_kernel<true, L>), dim3(blocks), dim3(L), 0, s, a);
    else
        hipLaunchKernelGGL((physics_kernel<false, L>), dim3(blocks), dim3(L), 0, s, a);
}

int launch_env_physics(v2p_env* env, hipStream_t s) {
    PhysArgs a;
    a.model = env->model->dev;
    a.state = env->state;
    a.ctrl = env->ctrl;
    a.out = env->out;
    a.ws = env->ws;
    a.contact_ids = env->contact_ids;
    a.contact_ids_sub = env->contact_ids_sub;
    a.prof = env->prof;
    a.par_pack[0] = a.par_pack[1] = a.ord_pack[0] = a.ord_pack[1] = 0ull;
    for (int i = 0; i < NB; ++i) {
        int par = env->model->host.parents[i] < 0 ? 0 : env->model->host.parents[i];
        a.par_pack[i / 12] |= (unsigned long long)par << (5 * (i % 12));
        a.ord_pack[i / 12] |= (unsigned long long)env->model->host.order[i] << (5 * (i % 12));
    }
    a.n = env->n;
    a.p = env->p;
    // environments per workgroup: 8 puts one wave on each of the 1024 SIMDs at 8192 envs (4 x 39 KB LDS per CU)
    static int envs_per_block = 0;
    if (!envs_per_block) {
        const char* s_env = debug_env("V2P_ENVS_PER_BLOCK");
        envs_per_block = s_env ? atoi(s_env) : 32;
    }
    const bool c = env->p.enable_contact != 0;
    switch (envs_per_block) {
        case 4: launch_L<4>(a, c, s); break;
        case 16: launch_L<16>(a, c, s); break;
        case 8: launch_L<8>(a, c, s); break;
        default: launch_L<32>(a, c, s); break;
    }
    return check_hip(hipGetLastError(), "physics_kernel");
}

}  // namespace v2p
