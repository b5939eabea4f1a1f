// Articulated rigid-body step, LANE = LINK mapping: 32 lanes per environment (24 links + 8 idle), two
// environments per wave64, every per-link quantity in registers.  Same model and same arithmetic per
// link as physics.hip (see that file and oracle/phys/v2p_phys_oracle.c for the model); what changes is
// the schedule:
//
//   * the three tree recursions run level-synchronously (max depth 8 instead of 23 sequential links);
//     parent -> child and child -> parent transfers are cross-lane pulls (ds_bpermute / DPP lane shifts) inside the wave
//   * contact generation: a box test culls, each link near the ground is scanned by a group of 8 lanes (DPP reductions), the
//     8 groups of a wave take the near links of both environments
//   * the block Gauss-Seidel sweep visits, per environment, only the bodies that environment touches
//     (k-th touched body of both environments at once) and keeps their velocities current by WALKING the tree from one
//     touched link to the next (up to the lowest common ancestor, which answers with its Lambda, and down again); one
//     root -> leaves pass at the end of the sweep moves every link (see the sweep)
//   * envs are handed to waves in descending order of their contact load (the heaviest quarter each next to one of the lightest),
//     a counting sort spread over this kernel's epilogue (bin histogram + arrival lists) and the next launch's prologue (every wave
//     looks its two envs up: slot -> rank -> bin -> arrival list); an env's arithmetic never depends on
//     the env it shares a wave with
//   * one launch = (substep, env pair) JOBS: the heaviest pairs run their four substeps in one workgroup, the others hand the
//     state over from job to job through memory (16-byte write-through stores / loads + a progress word per pair)
//   * pre-physics (PD-target clamp, residual wrench) runs in the prologue of an env's first job, post-physics (observation, reward,
//     reset flags, next target) in the epilogue of its last: v2p_env_step is this one kernel
//   * no global workspace: a job reads the state once, keeps it in registers and writes the state (and, the last job, the
//     caller's row-major tensors: lane = body gives contiguous rows) once; hull vertices and per-link constants come from the
//     (per-env) shape table through L1/L2; LDS holds the contact records of a link (28 floats per lane) and values a phase does not touch
//   * 168 VGPRs, 3 waves per SIMD (needs -fno-slp-vectorize: SLP packing costs ~160 registers here)
//
// The sequential semantics of the Gauss-Seidel sweep (bodies ascending; inside a body the limit rows of its joint, then its points in slot
// order, rows n, t1, t2) are those of the one-env-per-lane kernel and of the float64 oracle, so the results agree with both to rounding.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

// This translation unit is compiled with -fassociative-math -freciprocal-math -fno-signed-zeros -fno-trapping-math -fno-honor-nans
// (build.py) and NOT with -ffast-math: reassociation and NaN-free selects for the code of this file, but the device libraries it
// links and the way the HIP headers map sinf / cosf / atan2f stay the precise ones (-ffast-math, or the full set of its component
// flags, turns them into the hardware approximations for the whole file, which no pragma undoes; the helpers that restate the
// reference in the pre-physics prologue must round like task_ops.hip).  The physics model is this engine's own specification and
// is checked against the float64 oracle at fixed tolerances; its transcendental / division shortcuts are written out below
// (hardware sin, cos, sqrt, rcp).  V2P_LL_STRICT_MATH builds everything precise.
// (V2P_LL_PRECISE_SINCOS / _SQRT / _RCP switch one shortcut back at a time: tools/parity_ab.sh bisects a parity difference with them)
#if !defined(V2P_LL_STRICT_MATH)
#if !defined(V2P_LL_PRECISE_SINCOS)
#define PHYS_SINCOS(x, s, c) do { (s) = __sinf(x); (c) = __cosf(x); } while (0)
#endif
#if !defined(V2P_LL_PRECISE_SQRT)
#define PHYS_SQRT(x) __builtin_amdgcn_sqrtf(x)
#endif
#if !defined(V2P_LL_PRECISE_RCP)
#define PHYS_RCP(x) __builtin_amdgcn_rcpf(x)
#endif
#endif
#include "phys_common.hpp"
#include "hull_gjk.hpp"

namespace v2p {

// Pre-physics (humanoid_smpl_im.py:125-157, 391-396) restates the reference's torch arithmetic: PD-target clamp, residual root wrench
// rotated into the heading frame.  ONE implementation, compiled here with precise semantics and without contraction (the pragma
// below; this file is built -ffp-contract=fast-honor-pragmas), serves both the stand-alone env_pre_kernel and the prologue of the
// physics kernel, so the fused step equals the staged step bit for bit (tests).
namespace strict {
#pragma clang fp reassociate(off) reciprocal(off) contract(off)
#include "v2p_math.inc"
#include "motion_sample.inc"
#include "post_ops.inc"
// sum of 24 consecutive floats, ascending (the order env_post_kernel adds the bodies' reward terms in)
__device__ __forceinline__ float sum_bodies(const float* p) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i) s += p[i];
    return s;
}
__device__ __forceinline__ float pd_clamp(float act, float q, float lim) { return fmaxf(fminf(act, q + lim), q - lim); }
// root_rot: rigid-body rotation of the root (xyzw); a3: the three action components of the force (or torque) part
__device__ __forceinline__ V3 residual_wrench(const float* root_rot, float a0, float a1, float a2, float scale) {
    const Q4 hq = ref_heading_quat(ref_calc_heading(ref_remove_base_rot(Q4{root_rot[0], root_rot[1], root_rot[2], root_rot[3]})));
    return ref_quat_rotate(hq, V3{a0 * scale, a1 * scale, a2 * scale});
}
// force and torque at once: one heading quaternion (the same functions of the same arguments: the same bits as two calls)
__device__ __forceinline__ void residual_wrench2(const float* root_rot, const float* a6, float fscale, float tscale, V3& F, V3& T) {
    const Q4 hq = ref_heading_quat(ref_calc_heading(ref_remove_base_rot(Q4{root_rot[0], root_rot[1], root_rot[2], root_rot[3]})));
    F = ref_quat_rotate(hq, V3{a6[0] * fscale, a6[1] * fscale, a6[2] * fscale});
    T = ref_quat_rotate(hq, V3{a6[3] * tscale, a6[4] * tscale, a6[5] * tscale});
}
}  // namespace strict
#if !defined(V2P_LL_STRICT_MATH)
#pragma clang fp reassociate(on) reciprocal(on) contract(fast)  // (a file-scope fp pragma stays in force past the namespace: switch back)
#endif

typedef volatile __attribute__((address_space(3))) float lds_vfloat;  // a volatile float in LDS (ds_read / ds_write, 32-bit address + immediate offset)
constexpr int LPE = 32;  // lanes per environment

// wave-wide "any lane": a compare of the ballot in scalar registers (HIP's __any materialises the predicate as a 0/1 VGPR first:
// 2 VALU instructions per use, and the sweep loops use it ~20 times per block update)
__device__ __forceinline__ bool any64(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// hand-off of a job's results to another workgroup (possibly on another XCD: per-XCD L2s are not coherent, a CU's L1 is never refreshed):
// system-scope relaxed atomics = `sc0 sc1` loads / stores on both sides (MI355X_MICROARCH.md, inter-workgroup visibility, valid forms)
__device__ __forceinline__ uint8_t flag_ld(const uint8_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void flag_st(uint8_t* p, uint8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float cload(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void cstore(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// the same in 16-byte pieces (MI355X_MICROARCH.md: a dword `sc1` store is its own fabric write, ~6x the time per byte of a dwordx4 one;
// __hip_atomic_* stops at 8 bytes).  The compiler does not count these accesses: the loads are drained inside the asm block, the stores
// by the s_waitcnt vmcnt(0) in front of the progress word (memory operations retire in order, so the compiler's own counts only over-wait).
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cstore4(float* p, float x, float y, float z, float w) {
    const f4 v{x, y, z, w};
    // (s_nop 1: gfx940+ needs two wait states between a VMEM store of more than 8 bytes and a VALU write of its data registers; the
    // compiler's hazard recogniser does not look into inline asm and reuses the registers at once - with one wait state short, the
    // hand-offs of a loaded GPU carried the NEXT values of those registers)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void cload4x2(const float* p0, const float* p1, f4& a, f4& b) {
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p0), "v"(p1) : "memory");
}

// five chunks in one block (one wait for all of them): the whole hand-over of a lane
__device__ __forceinline__ void cload4x5(const float* p0, const float* p1, const float* p2, const float* p3, const float* p4, f4& a, f4& b, f4& c, f4& d, f4& e) {
    asm volatile("global_load_dwordx4 %0, %5, off sc0 sc1\n\tglobal_load_dwordx4 %1, %6, off sc0 sc1\n\tglobal_load_dwordx4 %2, %7, off sc0 sc1\n\t"
                 "global_load_dwordx4 %3, %8, off sc0 sc1\n\tglobal_load_dwordx4 %4, %9, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4) : "memory");
}

__device__ __forceinline__ float pull(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ V3 pull(V3 v, int s) { return V3{pull(v.x, s), pull(v.y, s), pull(v.z, s)}; }
__device__ __forceinline__ Q4 pull(Q4 q, int s) { return Q4{pull(q.x, s), pull(q.y, s), pull(q.z, s), pull(q.w, s)}; }
__device__ __forceinline__ Sym3 pull(const Sym3& a, int s) {
    return Sym3{pull(a.xx, s), pull(a.xy, s), pull(a.xz, s), pull(a.yy, s), pull(a.yz, s), pull(a.zz, s)};
}
__device__ __forceinline__ M3 pull(const M3& a, int s) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = pull(a.m[i], s);
    return r;
}
// lane i <- lane i+1 through the DPP network (wave_shl:1, VALU speed, no LDS round trip): links are in depth-first order, so the
// first child of any link is the next lane.  (The mirror image for parents, wave_shr:1 + a select against the non-chain links,
// measured slower than a plain ds_bpermute and is not used.)
// (bound_ctrl: the lane without a source - lane 63 - gets 0 from the hardware; with bound_ctrl off it KEEPS the `old` operand, which the
// compiler then has to materialise with a v_mov 0 in front of every one of these moves, and which keeps it from folding the move into the
// add that consumes it)
#ifndef V2P_LL_DPP_BOUND_CTRL
#define V2P_LL_DPP_BOUND_CTRL 1
#endif
__device__ __forceinline__ float from_next(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, V2P_LL_DPP_BOUND_CTRL != 0)); }
__device__ __forceinline__ V3 from_next(V3 v) { return V3{from_next(v.x), from_next(v.y), from_next(v.z)}; }
__device__ __forceinline__ Sym3 from_next(const Sym3& a) {
    return Sym3{from_next(a.xx), from_next(a.xy), from_next(a.xz), from_next(a.yy), from_next(a.yz), from_next(a.zz)};
}
__device__ __forceinline__ M3 from_next(const M3& a) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = from_next(a.m[i]);
    return r;
}
// lane i <- lane i-1 through the DPP network (wave_shr:1): the parent of a link that directly follows it
__device__ __forceinline__ float from_prev(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false)); }
#ifndef V2P_LL_DPP_PARENT
#define V2P_LL_DPP_PARENT 0
#endif
// value held by the parent lane.  `nonchain` = some link of the level being processed does not directly follow its parent: then the
// pull goes through ds_bpermute (an LDS round trip, ~100+ cycles of exposed latency per level for a wave whose partner is stalled
// too); on chain-only levels (V2P_LL_DPP_PARENT) it is a DPP shift, a VALU-speed move.  The branch is wave-uniform.
struct ParentPull {
    int plane;
    bool chain;
    template <typename F>
    __device__ __forceinline__ static void each(float* dst, const float* src, int n, F f) {
#pragma unroll
        for (int i = 0; i < n; ++i) dst[i] = f(src[i]);
    }
    __device__ __forceinline__ float operator()(float v, bool nonchain) const {
        if (V2P_LL_DPP_PARENT && !nonchain) return from_prev(v);
        return pull(v, plane);
    }
    // DPP on chain-only levels whatever V2P_LL_DPP_PARENT says (the per-update propagation of the sweep, V2P_LL_DPP_DOWN)
    __device__ __forceinline__ V3 fast(V3 v, bool nonchain) const {
        if (!nonchain) return V3{from_prev(v.x), from_prev(v.y), from_prev(v.z)};
        return V3{pull(v.x, plane), pull(v.y, plane), pull(v.z, plane)};
    }
    __device__ __forceinline__ V3 operator()(V3 v, bool nonchain) const {
        if (V2P_LL_DPP_PARENT && !nonchain) return V3{from_prev(v.x), from_prev(v.y), from_prev(v.z)};
        return V3{pull(v.x, plane), pull(v.y, plane), pull(v.z, plane)};
    }
    __device__ __forceinline__ Q4 operator()(Q4 q, bool nonchain) const {
        if (V2P_LL_DPP_PARENT && !nonchain) return Q4{from_prev(q.x), from_prev(q.y), from_prev(q.z), from_prev(q.w)};
        return Q4{pull(q.x, plane), pull(q.y, plane), pull(q.z, plane), pull(q.w, plane)};
    }
    __device__ __forceinline__ Sym3 operator()(const Sym3& a, bool nonchain) const {
        if (V2P_LL_DPP_PARENT && !nonchain) return Sym3{from_prev(a.xx), from_prev(a.xy), from_prev(a.xz), from_prev(a.yy), from_prev(a.yz), from_prev(a.zz)};
        return Sym3{pull(a.xx, plane), pull(a.xy, plane), pull(a.xz, plane), pull(a.yy, plane), pull(a.yz, plane), pull(a.zz, plane)};
    }
    __device__ __forceinline__ M3 operator()(const M3& a, bool nonchain) const {
        M3 r;
        if (V2P_LL_DPP_PARENT && !nonchain) {
#pragma unroll
            for (int i = 0; i < 9; ++i) r.m[i] = from_prev(a.m[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) r.m[i] = pull(a.m[i], plane);
        }
        return r;
    }
};
__device__ __forceinline__ Sym3 operator+(const Sym3& a, const Sym3& b) {
    return Sym3{a.xx + b.xx, a.xy + b.xy, a.xz + b.xz, a.yy + b.yy, a.yz + b.yz, a.zz + b.zz};
}
__device__ __forceinline__ Sym3 mask(bool c, const Sym3& a) {
    return Sym3{c ? a.xx : 0.f, c ? a.xy : 0.f, c ? a.xz : 0.f, c ? a.yy : 0.f, c ? a.yz : 0.f, c ? a.zz : 0.f};
}
__device__ __forceinline__ V3 mask(bool c, V3 a) { return V3{c ? a.x : 0.f, c ? a.y : 0.f, c ? a.z : 0.f}; }

// 8-lane group exchanges on the DPP network: lane ^ 1, lane ^ 2 (quad permutes) and lane -> 7 - lane (row_half_mirror)
__device__ __forceinline__ unsigned grp_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned grp_xor2(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned grp_mirror(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false); }
// (value, index) reductions over a group; index -1 = empty; ties go to the lower index so the result is what a serial scan finds
template <bool MAX, typename F>
__device__ __forceinline__ void grp_arg_step(float& v, int& k, F xch) {
    const float ov = __uint_as_float(xch(__float_as_uint(v)));
    const int ok = (int)xch((unsigned)k);
    const bool beats = MAX ? ov > v : ov < v;
    const bool take = ok >= 0 && (k < 0 || beats || (ov == v && ok < k));
    v = take ? ov : v;
    k = take ? ok : k;
}
__device__ __forceinline__ void grp_argmin(float& v, int& k) {
    grp_arg_step<false>(v, k, grp_xor1); grp_arg_step<false>(v, k, grp_xor2); grp_arg_step<false>(v, k, grp_mirror);
}
__device__ __forceinline__ void grp_argmax(float& v, int& k) {
    grp_arg_step<true>(v, k, grp_xor1); grp_arg_step<true>(v, k, grp_xor2); grp_arg_step<true>(v, k, grp_mirror);
}

#define LLSUB(k) do { if (DIAG && a.prof) { long long t_ = clock64(); if (((a.prof_heavy ? blockIdx.x < 8u : (blockIdx.x & 63) == 0)) && lane == 0) atomicAdd((unsigned long long*)&a.prof[k], (unsigned long long)(t_ - tsub)); tsub = t_; } } while (0)
#define LLPH(k) do { if (DIAG && a.prof) { long long t_ = clock64(); if (((a.prof_heavy ? blockIdx.x < 8u : (blockIdx.x & 63) == 0)) && lane == 0) atomicAdd((unsigned long long*)&a.prof[k], (unsigned long long)(t_ - tprev)); tprev = t_; } } while (0)

// (V2P_LL_WPB, waves per workgroup, is defined in v2p_internal.hpp: the host sizes the progress words with it)
#ifndef V2P_LL_WPS
#define V2P_LL_WPS 3   // waves per SIMD the register budget is set for (168 VGPRs; 2 = 256 VGPRs with everything in registers)
#endif
#ifndef V2P_LL_WPS_BALL
#define V2P_LL_WPS_BALL 3     // ... of the racket + ball instantiations
#endif
#ifndef V2P_LL_WPS_LIMITS
#define V2P_LL_WPS_LIMITS 3   // ... of the joint-limit instantiations without a ball
#endif
constexpr int LL_WPB = V2P_LL_WPB;
// V2P_LL_PARK2: phase-scoped parking of values that a phase does not touch (link velocities during pass 2 / contact generation / the
// Lambda recursion, Lambda of the root during contact generation, the contact records during the Lambda recursion): lowers the
// register peak of those phases so that nothing long-lived is spilled ACROSS the sweep loops when the kernel is built for 3 waves/SIMD.
#ifndef V2P_LL_PARK2
#define V2P_LL_PARK2 1
#endif
constexpr bool PARK2 = V2P_LL_PARK2 != 0;
// V2P_LL_PARK3: the contact records of a link (4 x offset, bias / gap, 3 impulses = 28 floats) live in the lane's LDS column instead of
// registers: they are touched once per block update (by the one lane being solved), not inside the per-level loops, and leave their
// 28 registers to the values those loops use.
#ifndef V2P_LL_PARK3
#define V2P_LL_PARK3 1
#endif
constexpr bool PARK3 = V2P_LL_PARK3 != 0;
#ifndef V2P_LL_KIN_JUMP
#define V2P_LL_KIN_JUMP 1     // pass 1 (kinematics) by pointer doubling over the tree instead of level by level
#endif
#ifndef V2P_LL_PREFETCH_ROWS
#define V2P_LL_PREFETCH_ROWS 1
#endif
#ifndef V2P_LL_FAST_HANDOVER
#define V2P_LL_FAST_HANDOVER 0  // prologue of a job that takes a hand-over: hand-over slots indexed by WAVE SLOT (not by env), every chunk of a lane in ONE
                                // load block, the env looked up from the pairing tables while the first poll of the progress word is in flight
                                // (0, the default: progress word -> env index -> chunks 0,1 -> chunks 48,49 -> chunk 54, up to five dependent round
                                // trips.  Measured round 4, profiles/r04_ab_handover_targethead.txt: no difference at 8192 / 32768 envs, racket + ball,
                                // TGS - with three waves per SIMD a job's prologue latency is covered by the other waves; the shorter chain is kept
                                // as a build switch, bit-identical to the default in the substep-job tests)
#endif
#ifndef V2P_LL_WAIT_SLEEP
#define V2P_LL_WAIT_SLEEP 16  // s_sleep argument (x 64 clocks) between two polls of a job that waits for its predecessor's hand-over
#endif
#ifndef V2P_LL_TARGET_HEAD
#define V2P_LL_TARGET_HEAD 0  // 1: fused step, the next target is sampled by the job of an env's FIRST substep (it depends on the clip and the time
                              // only) instead of by its last one, so that the jobs that end a launch get shorter.  Measured round 4 (same file):
                              // 19.89 vs 19.92 M at 8192 envs, 12.74 vs 12.86 M racket + ball, 24.48 vs 24.35 M at 32768: nothing - the tail of a
                              // launch is set by when the last jobs START, not by the ~8 us they lose
#endif
#ifndef V2P_LL_WALK
#define V2P_LL_WALK 1  // 0: the sweep with a leaf -> root -> leaves propagation after every touched group (A/B; the ball / joint-limit kernels use it)
#endif
#ifndef V2P_LL_ALT_SWEEP
#define V2P_LL_ALT_SWEEP 0  // 1 = experiment of round 4: PGS sweeps alternate their direction over the touched links (oracle: v2p_oracle_experiment(4)),
                            // the walk goes back and forth and never returns from the last link to the first: 8 % fewer instructions, +3.5 % at
                            // 8192 envs, +6 % at 32768 - and 1.5 x the distance to the converged solution after 4 sweeps (the link a sweep ends on is
                            // solved twice in a row): the gain is paid with solver accuracy, so it is NOT the model.  Parity of
                            // the switched build with the switched oracle was green on all 110 GPU tests (profiles/r04f_*).
#endif
#ifndef V2P_LL_DPP_DOWN
#define V2P_LL_DPP_DOWN 0
#endif
#ifndef V2P_LL_EXP
#define V2P_LL_EXP 0   // TIMING experiments of round 6 (tools/mkvariant.sh; bits 1, 2, 4 break the physics on purpose - they measure what a part costs):
                       // 1 no post-bounce after limit rows, 2 limit rows never change anything, 4 limit stops taken out of the walk (Lambda depth kept),
                       // 8 Lambda recursion only as deep as the CONTACT stops, 16 no joint ever has an active limit row (the code stays)
#endif
constexpr int PARK_TAR = 0, PARK_W0 = 3, PARK_XD0 = 6, PARK_Q = 9, PARK_X = 13, PARK_CR = 16, PARK_CB = 28, PARK_CL = 32,
              PARK_SCR = PARK3 ? 44 : 16,  // (one row of 64 dwords per wave: lane of the k-th near link)
              PARK_SLOTS = PARK_SCR + 1;  // LDS parking slots (dwords per lane)
constexpr int ROOTLAM_FLOATS = 2 * 24;  // (PARK2) Lambda of the two root links while the contacts are generated
// ball block (80 floats per env, after the parking area of the wave): state 13 | aero force 3 | ground contact: active gap bias lambda3 |
// point j at BL_RK + 16 j: active gap bias rl3 n3 lambda3 link (j = 0, 1: against the racket's cylinders; j = 2 .. 4: against the hulls
// of up to three links, RK_LINK = the link that owns the point) | velocity at the start of the substep 3
// (TGS: GGAP / RK_GAP advance slice by slice; BL_GREST / RK_REST = the restitution target of the row, rest x approach speed of v*, taken once
// at the start of the substep, a large value when the row does not bounce: it caps the bias of every slice)
constexpr int BL_POS = 0, BL_QUAT = 3, BL_VEL = 7, BL_ANG = 10, BL_F = 13, BL_GA = 16, BL_GGAP = 17, BL_GBIAS = 18, BL_GLAM = 19, BL_GREST = 22, BL_RK = 24,
              RK_A = 0, RK_GAP = 1, RK_BIAS = 2, RK_RL = 3, RK_N = 6, RK_LAM = 9, RK_LINK = 12, RK_REST = 13, NBREC = 5, BL_V0 = BL_RK + 16 * NBREC, BL_SLOTS = BL_V0 + 8;
constexpr int LDS_FLOATS_PER_WAVE = PARK_SLOTS * 64 + 2 * BL_SLOTS + ROOTLAM_FLOATS;
// ball x hull narrow phase, out of line: it runs on the few substeps in which a ball is within reach of a link, and inlined its
// registers would be spilled around on every substep
// (the closest point comes back BY VALUE, xyz = point, w = distance: an out-parameter by reference is a stack slot of the caller that the
// callee writes through a pointer)
#ifndef V2P_LL_GJK_INLINE
#define V2P_LL_GJK_INLINE 0
#endif
#if V2P_LL_GJK_INLINE
__device__ __forceinline__
#else
__device__ __noinline__
#endif
f4 ball_hull_distance(ConstShape* S, int v0, int nv, V3 c) {
    V3 p;
    const float d = hull_closest([&](int k) { return V3{S->hull_verts[v0 + k][0], S->hull_verts[v0 + k][1], S->hull_verts[v0 + k][2]}; }, nv, c, p);
    return f4{p.x, p.y, p.z, d};
}
// contact records of this lane's link: registers, or (PARK3) the lane's LDS column
template <bool LDS>
struct ContactStore;
template <>
struct ContactStore<false> {
    V3 r_[4];
    float b_[4];
    V3 l_[4];
    __device__ __forceinline__ explicit ContactStore(float*) {}
    __device__ __forceinline__ V3 cr(int c) const { return r_[c]; }
    __device__ __forceinline__ void set_cr(int c, V3 v) { r_[c] = v; }
    __device__ __forceinline__ float bias(int c) const { return b_[c]; }
    __device__ __forceinline__ void set_bias(int c, float v) { b_[c] = v; }
    __device__ __forceinline__ V3 lam(int c) const { return l_[c]; }
    __device__ __forceinline__ void set_lam(int c, V3 v) { l_[c] = v; }
};
template <>
struct ContactStore<true> {
    float* p;  // slot 0 of this lane's column
    __device__ __forceinline__ explicit ContactStore(float* park) : p(park) {}
    __device__ __forceinline__ V3 cr(int c) const { return V3{p[(PARK_CR + 3 * c) * 64], p[(PARK_CR + 3 * c + 1) * 64], p[(PARK_CR + 3 * c + 2) * 64]}; }
    __device__ __forceinline__ void set_cr(int c, V3 v) { p[(PARK_CR + 3 * c) * 64] = v.x; p[(PARK_CR + 3 * c + 1) * 64] = v.y; p[(PARK_CR + 3 * c + 2) * 64] = v.z; }
    __device__ __forceinline__ float bias(int c) const { return p[(PARK_CB + c) * 64]; }
    __device__ __forceinline__ void set_bias(int c, float v) { p[(PARK_CB + c) * 64] = v; }
    __device__ __forceinline__ V3 lam(int c) const { return V3{p[(PARK_CL + 3 * c) * 64], p[(PARK_CL + 3 * c + 1) * 64], p[(PARK_CL + 3 * c + 2) * 64]}; }
    __device__ __forceinline__ void set_lam(int c, V3 v) { p[(PARK_CL + 3 * c) * 64] = v.x; p[(PARK_CL + 3 * c + 1) * 64] = v.y; p[(PARK_CL + 3 * c + 2) * 64] = v.z; }
};

// friction frame of a hull x ground point from the link's v* (w0, xd0) and the point's offset rr: t1 along the tangential velocity of the
// point, t2 = z x t1; below 1e-6 m/s the world frame (oracle: v2p_oparams.friction_frame = 1)
__device__ __forceinline__ void vfric_frame(const V3& w0, const V3& xd0, const V3& rr, V3& t1, V3& t2) {
    const float vx = xd0.x + w0.y * rr.z - w0.z * rr.y, vy = xd0.y + w0.z * rr.x - w0.x * rr.z;
    const float s2 = vx * vx + vy * vy;
    const bool on = s2 > 1e-12f;
    const float inv = on ? rsqrtf(s2) : 0.f;
    t1 = on ? V3{vx * inv, vy * inv, 0.f} : V3{1.f, 0.f, 0.f};
    t2 = V3{-t1.y, t1.x, 0.f};
}
// TGS: temporal Gauss-Seidel with frozen Jacobians (v2p_sim_cfg.solver_type 1; the model is stated in oracle/phys/v2p_phys_oracle.c):
// cbias[] then holds the GAP of each point, advanced after every sweep, and the row bias is evaluated where it is used.
// DIAG: the per-phase cycle counters (V2P_PHASE_TIMING) and per-wave timeline stamps (V2P_WAVE_TIMES) are compiled into a separate
// instantiation: in the production kernel they cost registers (spills) and ~2 scalar instructions per probe inside the sweep loops.
// BALL: racket + ball: the first idle lane of an env (lb == 24) simulates the free ball, its state and the ball contact
// records live in a 64-float LDS block per env; the ball-racket rows are solved inside the block update of the racket's link.
// JOBS: one launch = substeps x waves JOBS.  Job (s, p) runs substep s of wave p's env pair and hands the state over to job (s+1, p)
// through global memory (system-scope stores / loads + a progress word per pair).  With whole control steps as jobs, 8192 envs are
// 4096 indivisible jobs of 0.14-0.65 ms on 2048-3072 wave slots and the launch is as long as its worst slot (82 % utilisation,
// profiles/r02*_wave_times.txt); quarter-size jobs pack 4x finer.  Jobs are numbered (and, as observed, dispatched) substep-major,
// heavy pairs first, so a job normally waits for one that started before it; a job whose predecessor does not show up within the
// time-out RECOMPUTES the pair's earlier substeps itself (see the wait), so neither progress nor results depend on the dispatch order.
// (Numbering the jobs by tickets drawn from an atomic counter as the workgroups start - progress by construction - measured -13 %:
// 13 k atomics on one address per launch.)
// LIMITS: joint-limit rows (v2p_sim_cfg.joint_limits; the model is stated in oracle/phys/v2p_phys_oracle.c).  A DOF whose range is
// narrower than a full turn carries one row against its nearer limit; the rows of joint b form their own block update right before
// the contact block of link b.  A limit impulse is a joint-space impulse: it enters the propagation as the link's `un`, and its
// reaction (-impulse, a pure torque) joins what the link hands up to its parent.  The inverse mass of the rows is the joint-space
// inverse inertia K = Di + ((T - 1)^T G (T - 1))_ww = Lambda_b,ww - H1 - H1^T + Lambda_parent,ww in the notation of the recursion below.
// VFRIC: v2p_sim_cfg.friction_frame = velocity (ABI 14; the model is stated in oracle/phys/v2p_phys_oracle.c).  The tangent rows of a hull x
// ground point lie along / across the tangential velocity the point has under v* (the link's velocity at the start of the sweep, which
// stays parked in the PARK_W0 / PARK_XD0 slots of the lane's LDS column for the whole sweep) instead of along world x / y: the frame is
// re-derived from those six floats at every visit of the link (one rsqrt per point) - eight more floats per lane of LDS to keep it would
// cost the third wave per SIMD.  An instantiation of its own: with general tangent directions the rows lose the zeros the world frame
// gives them (~+40 % instructions in the friction rows), which the default kernel must not pay.
template <bool CONTACT, bool MULTI, bool TGS, bool DIAG, bool BALL, bool JOBS, bool LIMITS, bool VFRIC = false>
__global__ __launch_bounds__(64 * LL_WPB, DIAG ? 2 : (BALL ? V2P_LL_WPS_BALL : (LIMITS ? V2P_LL_WPS_LIMITS : V2P_LL_WPS))) void physics_ll_kernel(PhysArgs a) {
    constexpr bool WALK = V2P_LL_WALK != 0;  // the sweep as one walk over the tree (see the sweep)
    constexpr bool OPAQUE_H = BALL || LIMITS;  // (the headline instantiation keeps its 0 - 12 B of scratch either way: measured no difference, profiles/r06d_variants_spill.log)
    const int64_t N = a.n;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int lb = lane & (LPE - 1);
    const bool valid = lb < NB;
    const int b = valid ? lb : 0;  // idle lanes shadow link 0 and never commit anything
    const int base = lane & LPE;
    // JOBS: the first job_mono workgroups run ALL substeps of the job_mono heaviest env pairs (their chain of substeps is the critical
    // path of the launch: it starts at once and never waits); the other pairs are cut into one job per substep, substep-major
    const int nblk = JOBS ? a.job_blocks : (int)gridDim.x;                 // env pairs (wave slots) of the launch
    const int jid = (int)blockIdx.x;  // jobs are numbered substep-major, heavy pairs first
    const bool mono = !JOBS || jid < a.job_mono;
    const int jcut = JOBS && nblk > a.job_mono ? nblk - a.job_mono : 1;     // pairs that are cut into substep jobs
    const int jrel = JOBS ? jid - a.job_mono : 0;
    const int sjob = mono ? 0 : jrel / jcut;                               // the substep this job runs
    // Order of the cut pairs within a substep.  Wave slots 0 .. pl_mix - 1 hold a heavy env next to a light one, the slots behind them
    // pairs of equals in descending rank - and the first of THOSE (two mid-heavy envs: the union of two contact structures) cost as much
    // as the heavy x light slots in front of them, yet were dispatched behind all of them: they ended every launch (job timeline:
    // the last 15 % of a launch ran at half occupancy).  The two lists are dispatched alternately, so both kinds of long job start early
    // and the tail of a substep is made of short ones.  (Only the order of dispatch changes: results are bit-identical.)
    int jpair = jrel % jcut;
    if (JOBS && !mono && a.job_interleave) {
        const int nmix = a.pl_mix > a.job_mono ? (a.pl_mix < nblk ? a.pl_mix : nblk) - a.job_mono : 0;  // cut slots that are heavy x light
        const int neq = jcut - nmix, both = 2 * (nmix < neq ? nmix : neq);
        if (jpair < both) jpair = (jpair & 1) ? nmix + (jpair >> 1) : (jpair >> 1);
        else jpair = nmix < neq ? jpair : jpair - neq;   // the rest of the longer list (neq >= nmix: positions keep their index)
    }
    const int bid = mono ? jid : a.job_mono + jpair;
    const int64_t slot = ((int64_t)bid * LL_WPB + (threadIdx.x >> 6)) * 2 + half;
    const bool live_env = slot < N;
    const int jlen = JOBS ? a.job_len : 1;  // substeps per job (v2p_env: job_len; 1 = one job per substep)
    // first substep of job j of a cut pair: the FIRST job may be longer than the others (job_lead substeps: one hand-over less per pair,
    // while the jobs that end a launch keep the fine granularity)
    const int jlead = JOBS ? a.job_lead : 1;
    auto jstart = [&](int j) -> int { return j <= 0 ? 0 : jlead + (j - 1) * jlen; };
    const bool first_job = mono || sjob == 0, last_job = mono || jstart(sjob + 1) >= a.p.nsub;
    // the inputs of this job were written by another workgroup of this launch - unless that one did not show up in time (below)
    bool handed = !mono && sjob > 0;
    int* const progress = JOBS ? a.job_progress + (bid * LL_WPB + (threadIdx.x >> 6)) : nullptr;
    // (FAST_HANDOVER: the first poll of the progress word is issued HERE and looked at behind the env lookup below - the two round trips
    // of a job's prologue that do not depend on each other overlap)
    int poll0 = 0;
    if constexpr (JOBS && V2P_LL_FAST_HANDOVER != 0) if (handed && lane == 0) poll0 = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // A job of a cut pair that starts after the pair's step is COMPLETE - its successors gave up waiting, replayed its substeps and the
    // last of them has written the results - must not run: the inputs of the step (state, actions, reset flags) are already those of the
    // next one.  The job of an env's last substep leaves the word at launch x (nsub + 1) + nsub when it ends; every other job looks before
    // it touches anything.  (Forward progress never needs this; it keeps a pathologically late job from writing the exposed PD targets,
    // per-call ball records or hand-overs from the wrong inputs.)
    if constexpr (JOBS) if (!mono) {
        int done = 0;
        if (lane == 0) done = (handed && V2P_LL_FAST_HANDOVER != 0 ? poll0 : __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) >= a.job_epoch * (a.p.nsub + 1) + a.p.nsub;
        if (__builtin_amdgcn_readfirstlane(done)) {
            // (counted: what this job owns besides its substeps - the exposed PD targets, the in-place action masking, the ball's per-call
            // records - was published by nobody for that step; v2p_env_check reports it as an error, not as lost time)
            if (lane == 0 && half == 0) __hip_atomic_fetch_add(a.job_progress + a.job_blocks * LL_WPB + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
    auto wait_for_predecessor = [&]() {
    if constexpr (JOBS) if (handed) {
        // wait for the previous substep of this env pair
        // progress word of a pair = launch number x (nsub + 1) + substeps handed over: the last hand-over of launch E stores
        // E (nsub + 1) + nsub - 1, below every value a job of launch E + 1 waits for, whatever nsub is (vid2player's controller
        // configs run substeps 6 x controlFrequencyInv 2 = 12 per control step)
        const int want = a.job_epoch * (a.p.nsub + 1) + sjob;
        int ok = 1;
        if (lane == 0) {
            long spins = 0;
            while ((V2P_LL_FAST_HANDOVER != 0 && spins == 0 ? poll0 : __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) < want) {
                if (spins >= a.job_timeout_spins) {
                    // The predecessor is late beyond reason (jobs are dispatched in index order as far as observed, but nothing promises
                    // it).  This job then runs the pair's EARLIER substeps itself, from the inputs of the step, before its own: every job
                    // of a pair computes the same bits whoever runs it, the hand-over slots are per substep, and the replayed substeps
                    // publish nothing - so the late predecessor, when it does run, rewrites identical values and nothing else changes.
                    // Progress and results are independent of the dispatch order; only time is lost (counted: v2p_env_job_recoveries).
                    ok = 0;
                    __hip_atomic_fetch_add(a.job_progress + a.job_blocks * LL_WPB, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(V2P_LL_WAIT_SLEEP);
                ++spins;
            }
        }
        handed = __builtin_amdgcn_readfirstlane(ok) != 0;
        __builtin_amdgcn_wave_barrier();
    }
    };
    if constexpr (V2P_LL_FAST_HANDOVER == 0) wait_for_predecessor();
    int64_t e = live_env ? slot : N - 1;
    if (V2P_LL_FAST_HANDOVER == 0 && a.pl_start && handed) {
        // (looked up by the job of the pair's first substep, which has handed it over with everything else)
        e = __hip_atomic_load(&a.pl_slot_env[live_env ? slot : N - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (a.pl_start) {
        // ---- which env this wave slot simulates: slot -> rank by contact load (the heaviest `mix` envs sit in the even slots 0, 2, ...,
        // each next to one of the lightest in the odd slot; the rest follows by rank) -> load bin (the last of the 256 bins that starts at
        // or before the rank: one 16-byte load per lane + four ballots) -> the env that arrived in that bin as number (rank - start).
        // The tables are what the previous launch left (physics epilogue); looked up here, no kernel has to scatter them in between.
        const int nn = (int)N, mixn = a.pl_mix;
        const int64_t sl0 = ((int64_t)bid * LL_WPB + (threadIdx.x >> 6)) * 2;
        const int s0 = (int)(sl0 < N ? sl0 : N - 1), s1 = (int)(sl0 + 1 < N ? sl0 + 1 : N - 1);
        const int r0 = s0 < 2 * mixn ? ((s0 & 1) ? nn - 1 - (s0 >> 1) : (s0 >> 1)) : s0 - mixn;
        const int r1 = s1 < 2 * mixn ? ((s1 & 1) ? nn - 1 - (s1 >> 1) : (s1 >> 1)) : s1 - mixn;
        const int4 st = ((const int4*)a.pl_start)[lane];
        const int c0 = __popcll(__ballot(st.x <= r0)) + __popcll(__ballot(st.y <= r0)) + __popcll(__ballot(st.z <= r0)) + __popcll(__ballot(st.w <= r0));
        const int c1 = __popcll(__ballot(st.x <= r1)) + __popcll(__ballot(st.y <= r1)) + __popcll(__ballot(st.z <= r1)) + __popcll(__ballot(st.w <= r1));
        const int bin = half ? c1 - 1 : c0 - 1, rk = half ? r1 : r0;
        e = a.pl_list[(int64_t)bin * N + (rk - a.pl_start[bin])];
        if (V2P_LL_FAST_HANDOVER == 0 && JOBS && !mono && lb == 0 && live_env) __hip_atomic_store(&a.pl_slot_env[slot], (int32_t)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if constexpr (V2P_LL_FAST_HANDOVER != 0) wait_for_predecessor();
    // the env index behind an opaque move: addresses formed from it inside the substep loop are computed where they are used instead of
    // being hoisted in front of the loop and kept (spilled: 64-bit pointers, 8 bytes of scratch per lane each) across all of it
    auto env_here = [&]() -> int64_t { int64_t v = e; asm volatile("" : "+v"(v)); return v; };
    ConstModel& M = *(ConstModel*)a.model;
    // numeric data of this env's body shape: the model's own shape, or (MULTI) one of the batch's shapes, per env
    const int sid = MULTI ? a.env_shape[e] : 0;
    ConstShape* S = MULTI ? (ConstShape*)(a.shapes + sid) : &M.shape;
    float* __restrict__ st = a.state;
    const EnvParams& P = a.p;
    const float h_launch = P.h;
    const int maxd = M.max_depth;
    const int multi = M.multi_child_levels;
    const int nonchain = M.nonchain_levels;

    // hull vertex `idx` of this env's shape, from the shape table through L1/L2.  (An LDS copy per workgroup measured 3.5 % slower
    // at 8192 envs: 4096 workgroups x 19 KB of staging traffic and a barrier before the first substep.)
    auto hullv = [&](int idx) -> float4 { return make_float4(S->hull_verts[idx][0], S->hull_verts[idx][1], S->hull_verts[idx][2], 0.f); };

    // ---- LDS "parking": per-lane values that live for the whole launch but are touched once or twice per substep (the PD target,
    // the velocities at the start of the sweep).  Holding them in registers made the allocator spill them (or something else) to
    // scratch - private memory that ends up as HBM write traffic; 9 dwords x 64 lanes of LDS per wave cost nothing.
    extern __shared__ float park_all[];
    float* const park = park_all + (threadIdx.x >> 6) * LDS_FLOATS_PER_WAVE + lane;  // slot k of this lane: park[k * 64]
    // this env's ball block.  The pointer carries the LDS address space in its TYPE: the accesses are volatile (the ball lane writes what
    // link lanes read within the wave), and address-space inference leaves volatile accesses alone - as a plain `volatile float*` every
    // bl[k] was a FLAT load / store through its own 64-bit address, ~50 loop-invariant pointers that were kept (spilled: 400 B of
    // scratch per lane) across the substep loop
    lds_vfloat* const bl = (lds_vfloat*)(park_all + (threadIdx.x >> 6) * LDS_FLOATS_PER_WAVE + PARK_SLOTS * 64 + half * BL_SLOTS);
    auto park_put3 = [&](int slot, V3 v) { park[slot * 64] = v.x; park[(slot + 1) * 64] = v.y; park[(slot + 2) * 64] = v.z; };
    auto park_get3 = [&](int slot) -> V3 { return V3{park[slot * 64], park[(slot + 1) * 64], park[(slot + 2) * 64]}; };
    // (PARK2) the link velocities leave the registers for the phases that do not touch them; the W0 / XD0 slots are free outside the sweep
    auto park_vel = [&](const V3& w_, const V3& xd_) { if (PARK2) { park_put3(PARK_W0, w_); park_put3(PARK_XD0, xd_); } };
    auto unpark_vel = [&](V3& w_, V3& xd_) { if (PARK2) { w_ = park_get3(PARK_W0); xd_ = park_get3(PARK_XD0); } };

    // ---- per-lane model constants
    const int par = b ? M.parents[b] : 0;
    const int plane = base + par;
    const ParentPull pp{plane, par == b - 1};
    const bool firstchild = par == b - 1;  // depth-first order: the first child directly follows its parent
    const int dep = valid ? M.depth[b] : 99;
    const int c0 = M.children[b][0], c1 = M.children[b][1], c2 = M.children[b][2];
    const bool has0 = valid && c0 >= 0, has1 = valid && c1 >= 0, has2 = valid && c2 >= 0;
    const int cl1 = has1 ? base + c1 : lane, cl2 = has2 ? base + c2 : lane;
    const float aug = MULTI ? a.shape_aug[sid * NB + b] : P.aug[b];
    const int desc = M.desc_mask[b];  // links of the subtree rooted here (self included)
    const int aj = valid ? M.anc_jump[b] : -1;  // the ancestors 1, 2, 4, 8 levels up, 8 bits each (255 = none): the kinematics by pointer doubling
    const int jrounds = M.jump_rounds;
    // Kinematics of the whole tree in jrounds = ceil(log2(depth + 1)) rounds of pointer doubling (4 for the SMPL tree, depth 8) instead of
    // one step per level: every link holds its pose RELATIVE to the ancestor 2^k levels up (to the world once that ancestor lies above the
    // root) and composes it with that ancestor's entry - all lanes work in every round, where the level loop has the links of one depth
    // work and the others wait (~1100 -> ~420 instruction slots per substep).  Poses first, then the velocities the same way:
    //   w_b = w_A + W,   xd_b = xd_A + w_A x (x_b - x_A) + V       (W, V) of b relative to A;  composing with A's own (W_A, V_A) over A':
    //   W' = W_A + W,    V' = V_A + W_A x (x_b - x_A) + V
    // The root's lane holds the world values (the floating base's state) from the start.  qj / wj: the joint's quaternion and rate (body
    // axes), lp: the joint's offset in the parent's frame.  Out: world pose and velocity of every link, rr = the offset in world axes,
    // wrel = the joint's rate in world axes.
    auto kin_jump = [&](Q4& q_, V3& x_, V3& w_, V3& xd_, const Q4& qj, const V3& wj, const V3& lp, V3& rr, V3& wrel) {
        const bool link = valid && dep >= 1;
        Q4 Q = link ? qj : q_;
        V3 X = link ? lp : x_;
        for (int rd = 0; rd < jrounds; ++rd) {
            const int al = (aj >> (8 * rd)) & 255;
            const bool has = al != 255;
            const int src = has ? base + al : lane;
            const Q4 Qa = pull(Q, src);
            const V3 Xa = pull(X, src);
            if (has) {
                X = Xa + mul(q2mat(Qa), X);
                Q = qmul(Qa, Q);
            }
        }
        if (link) {
            q_ = qnormalize(Q);
            x_ = X;
        }
        const Q4 pq = pull(q_, plane);
        if (link) {
            rr = mul(q2mat(pq), lp);
            wrel = mul(q2mat(q_), wj);
        }
        V3 W = link ? wrel : w_, V = link ? V3{0.f, 0.f, 0.f} : xd_;
        for (int rd = 0; rd < jrounds; ++rd) {
            const int al = (aj >> (8 * rd)) & 255;
            const bool has = al != 255;
            const int src = has ? base + al : lane;
            const V3 Wa = pull(W, src), Va = pull(V, src), xa = pull(x_, src);
            if (has) {
                V = Va + cross(Wa, x_ - xa) + V;
                W = Wa + W;
            }
        }
        if (link) {
            w_ = W;
            xd_ = V;
        }
    };
    // opt-in (v2p_sim_cfg.freeze_terminated_envs): an env whose reset flag is set keeps its state; a wave whose two envs are frozen
    // skips the substeps altogether (frozen envs sort to the end of the launch order, so they share waves)
    const bool frozen = P.freeze_terminated && a.reset[e] == 1;
    const int nsub = (P.freeze_terminated && !any64(!frozen)) ? 0 : P.nsub;
#if defined(V2P_LL_PRIO_MONO)
    if (JOBS && mono) __builtin_amdgcn_s_setprio(V2P_LL_PRIO_MONO);  // the heaviest pairs are the critical path of the launch
#endif
#if defined(V2P_LL_PRIO)
    {   // issue priority by predicted load: the heaviest pairs are the critical path of the launch, light waves fill the gaps they leave
        const int k0 = a.pair_key ? a.pair_key[e] : 0;
        const int kmax = __builtin_amdgcn_readfirstlane(max(k0, __shfl_xor(k0, 32)));
        if (kmax >= 96) __builtin_amdgcn_s_setprio(3);
        else if (kmax >= 64) __builtin_amdgcn_s_setprio(2);
        else if (kmax >= 40) __builtin_amdgcn_s_setprio(1);
    }
#endif
    // substeps of this job (a job that gave up waiting replays the earlier ones)
    const int sub0 = handed ? jstart(sjob) : 0, sub1 = mono ? nsub : (nsub ? (jstart(sjob + 1) < nsub ? jstart(sjob + 1) : nsub) : 0);
    auto ldin = [&](const float* p) -> float { return handed ? cload(p) : *p; };

    // inputs of the fused pre-physics, REQUESTED here, in front of the hand-over loads (inline asm that waits for its own loads), and used
    // behind them: the two memory latencies of a job's prologue overlap
    const bool pre_on = a.actions && valid && live_env;
    long long pre_reset = 0;
    float pre_a0 = 0.f, pre_a1 = 0.f, pre_a2 = 0.f, pre_q0 = 0.f, pre_q1 = 0.f, pre_q2 = 0.f;
    if (pre_on) {
        pre_reset = a.reset[e];
        if (b != 0) {
            const float* ap = a.actions + e * NACT + 3 * (b - 1);
            const float* qd = a.x_dof + (e * NDOF + 3 * (b - 1)) * 2;
            pre_a0 = ap[0]; pre_a1 = ap[1]; pre_a2 = ap[2];
            pre_q0 = qd[0]; pre_q1 = qd[2]; pre_q2 = qd[4];
        }
    }
    // ---- state: the root lane carries the root pose/velocity, every other lane its joint
    Q4 q{0.f, 0.f, 0.f, 1.f}, jq{0.f, 0.f, 0.f, 1.f};
    V3 x{0.f, 0.f, 0.f}, w{0.f, 0.f, 0.f}, xd{0.f, 0.f, 0.f}, wt{0.f, 0.f, 0.f}, tar{0.f, 0.f, 0.f};
    // (JOBS: the job of an env's first substep reads the engine's state, the others what the job before them handed over: 16-byte chunks,
    // chunk 2b, 2b+1 = joint b (quaternion | rate), chunks 0, 1, 48, 49 = the root (quat | pos, vx | vy, vz, wx, wy | wz))
    // (one hand-over slot per substep: slot s holds the state after substep s)
    // (FAST_HANDOVER: the slot of a hand-over is the WAVE SLOT of the pair - known from the workgroup index alone - not the env)
    const int64_t hidx = V2P_LL_FAST_HANDOVER != 0 ? (live_env ? slot : N - 1) : e;
    float* const hand = JOBS && handed ? a.job_hand + ((int64_t)(sjob - 1) * N + hidx) * HAND_FLOATS : nullptr;
    bool got = false;
    bool bgot = false;  // (BALL) the ball lane has its state from the hand-over
    if constexpr (JOBS && V2P_LL_FAST_HANDOVER != 0) if (handed) {
        // every chunk of this lane in one block of five loads and ONE wait: a link lane its two chunks (the other three addresses repeat
        // the first), the root lane chunks 0, 1, 48, 49, 54, the ball lane chunks 50 .. 53
        const bool isball = BALL && lb == NB;
        const int k0 = isball ? 50 : 2 * b, k1 = isball ? 51 : 2 * b + 1, k2 = isball ? 52 : (b == 0 ? 48 : 2 * b), k3 = isball ? 53 : (b == 0 ? 49 : 2 * b),
                  k4 = (!isball && b == 0) ? 54 : k0;
        f4 c0, c1, c2, c3, c4;
        cload4x5(hand + 4 * k0, hand + 4 * k1, hand + 4 * k2, hand + 4 * k3, hand + 4 * k4, c0, c1, c2, c3, c4);
        if (isball) {
            lds_vfloat* const blh = (lds_vfloat*)(park_all + (threadIdx.x >> 6) * LDS_FLOATS_PER_WAVE + PARK_SLOTS * 64 + half * BL_SLOTS);
            blh[0] = c0.x; blh[1] = c0.y; blh[2] = c0.z; blh[3] = c0.w; blh[4] = c1.x; blh[5] = c1.y; blh[6] = c1.z; blh[7] = c1.w;
            blh[8] = c2.x; blh[9] = c2.y; blh[10] = c2.z; blh[11] = c2.w; blh[12] = c3.x; blh[13] = c3.y; blh[14] = c3.z; blh[15] = c3.w;
            bgot = true;
        } else if (b == 0) {
            q = Q4{c0.x, c0.y, c0.z, c0.w};
            x = V3{c1.x, c1.y, c1.z};
            xd = V3{c1.w, c2.x, c2.y};
            w = V3{c2.z, c2.w, c3.x};
            if (a.actions && valid && jstart(sjob) < a.p.hold_sub) {
                float* const wp = park_all + (threadIdx.x >> 6) * LDS_FLOATS_PER_WAVE + base;
                wp[PARK_TAR * 64] = c3.y; wp[(PARK_TAR + 1) * 64] = c3.z; wp[(PARK_TAR + 2) * 64] = c3.w;
                wp[25 + PARK_TAR * 64] = c4.x; wp[25 + (PARK_TAR + 1) * 64] = c4.y; wp[25 + (PARK_TAR + 2) * 64] = c4.z;
            }
        } else {
            jq = Q4{c0.x, c0.y, c0.z, c0.w};
            wt = V3{c1.x, c1.y, c1.z};
        }
        got = true;
    }
    if constexpr (JOBS && V2P_LL_FAST_HANDOVER == 0) if (handed) {
        f4 c0, c1;
        cload4x2(hand + 8 * b, hand + 8 * b + 4, c0, c1);
        if (b == 0) {
            f4 c2, c3;
            cload4x2(hand + 4 * 48, hand + 4 * 49, c2, c3);
            q = Q4{c0.x, c0.y, c0.z, c0.w};
            x = V3{c1.x, c1.y, c1.z};
            xd = V3{c1.w, c2.x, c2.y};
            w = V3{c2.z, c2.w, c3.x};
            if (a.actions && valid && jstart(sjob) < a.p.hold_sub) {
                // the residual wrench of the fused step travels with the root's chunks while a later job still needs it (it is held for the
                // first hold_sub substeps only): force behind w.z, torque in chunk 54
                f4 c4, c5;
                cload4x2(hand + 4 * 54, hand + 4 * 54, c4, c5);
                float* const wp = park_all + (threadIdx.x >> 6) * LDS_FLOATS_PER_WAVE + base;
                wp[PARK_TAR * 64] = c3.y; wp[(PARK_TAR + 1) * 64] = c3.z; wp[(PARK_TAR + 2) * 64] = c3.w;
                wp[25 + PARK_TAR * 64] = c4.x; wp[25 + (PARK_TAR + 1) * 64] = c4.y; wp[25 + (PARK_TAR + 2) * 64] = c4.z;
            }
        } else {
            jq = Q4{c0.x, c0.y, c0.z, c0.w};
            wt = V3{c1.x, c1.y, c1.z};
        }
        got = true;
    }
    if (b == 0) {
        if (!got) {
            q = Q4{st[SIDX(ST_ROOT_QUAT + 0)], st[SIDX(ST_ROOT_QUAT + 1)], st[SIDX(ST_ROOT_QUAT + 2)], st[SIDX(ST_ROOT_QUAT + 3)]};
            x = V3{st[SIDX(ST_ROOT_POS + 0)], st[SIDX(ST_ROOT_POS + 1)], st[SIDX(ST_ROOT_POS + 2)]};
            xd = V3{st[SIDX(ST_VEL + 0)], st[SIDX(ST_VEL + 1)], st[SIDX(ST_VEL + 2)]};
            w = V3{st[SIDX(ST_VEL + 3)], st[SIDX(ST_VEL + 4)], st[SIDX(ST_VEL + 5)]};
        }
    } else {
        const int jb = ST_JQUAT + 4 * (b - 1), vb = ST_VEL + 6 + 3 * (b - 1), cb = CT_PD + 3 * (b - 1);
        if (!got) {
            jq = Q4{st[SIDX(jb + 0)], st[SIDX(jb + 1)], st[SIDX(jb + 2)], st[SIDX(jb + 3)]};
            wt = V3{st[SIDX(vb + 0)], st[SIDX(vb + 1)], st[SIDX(vb + 2)]};
        }
        if (!a.actions) tar = V3{ldin(&a.ctrl[CIDX(cb + 0)]), ldin(&a.ctrl[CIDX(cb + 1)]), ldin(&a.ctrl[CIDX(cb + 2)])};
    }
    // the residual root wrench of the fused step: lanes 0 (force) and 25 (torque) of the env keep it in the PARK_TAR slots of their LDS columns
    // (neither has a joint target)
    float* const wrench_park = park_all + (threadIdx.x >> 6) * LDS_FLOATS_PER_WAVE + base;
    if (pre_on) {
        // ---- pre-physics fused in (same functions, same rounding as env_pre_kernel below): lane b owns the three action components of
        // its joint, the root lane the residual wrench; dead envs are masked in place on the caller's tensor.  EVERY job of the env derives
        // the PD targets and the wrench itself - from the caller's actions and the exposed state of the start of the step, which nothing
        // overwrites before the env's last job has read them - instead of having the first job hand them over through `ctrl`: 75 dword
        // write-through stores per env, each its own fabric write (PMC: WRITE_SIZE)
        const bool dead = pre_reset == 1;
        if (b != 0) {
            float* ap = a.actions + e * NACT + 3 * (b - 1);
            float ax = pre_a0, ay = pre_a1, az = pre_a2;
            if (dead) {
                ax = ay = az = 0.f;
                if (first_job) { ap[0] = 0.f; ap[1] = 0.f; ap[2] = 0.f; }
            }
            const float lim = P.pd_tar_lim;
            tar = V3{strict::pd_clamp(ax, pre_q0, lim), strict::pd_clamp(ay, pre_q1, lim), strict::pd_clamp(az, pre_q2, lim)};
            if (first_job) {
                float* pt = a.pd_target + e * NDOF + 3 * (b - 1);
                pt[0] = tar.x; pt[1] = tar.y; pt[2] = tar.z;
            }
        } else {
            float* ap = a.actions + e * NACT + NDOF;
            float af[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const bool need_wrench = sub0 < P.hold_sub && !handed;
            if (!dead && need_wrench) {
#pragma unroll
                for (int k = 0; k < 6; ++k) af[k] = ap[k];
            }
            if (dead && first_job) {
#pragma unroll
                for (int k = 0; k < 6; ++k) ap[k] = 0.f;
            }
            if (need_wrench) {  // (a job past the substeps that hold the wrench has no use for it; a later job is handed it)
                const float* rq4 = a.x_rb + e * NB * 13 + 3;
                strict::V3 F, Tq;
                strict::residual_wrench2(rq4, af, P.res_force_scale, P.res_torque_scale, F, Tq);
                wrench_park[PARK_TAR * 64] = F.x; wrench_park[(PARK_TAR + 1) * 64] = F.y; wrench_park[(PARK_TAR + 2) * 64] = F.z;
                wrench_park[25 + PARK_TAR * 64] = Tq.x; wrench_park[25 + (PARK_TAR + 1) * 64] = Tq.y; wrench_park[25 + (PARK_TAR + 2) * 64] = Tq.z;
            }
        }
    }

    if constexpr (JOBS && !DIAG && V2P_LL_TARGET_HEAD != 0) {
        // ---- the NEXT target (reference state one step ahead of the new time), sampled here, in the job of the env's first substep: it
        // depends on the clip and the time only.  Same function, same arguments as the staged post-physics: the same bits.  (target[cur] -
        // what the reward of this step compares with - is not touched; the buffer written here held the target before it.)
        if (a.post.on && first_job && valid && live_env) {
            const PostArgs& Z = a.post;
            strict::post_sample_target(Z.b, Z.t, P, Z.motion_id[e], Z.b.cur_time[e] + P.dt, Z.cur, e, b);
        }
    }
    if (valid && b != 0) park_put3(PARK_TAR, tar);  // constant for the whole launch (the columns of lanes 0 and 25 hold the wrench there)
    const BallDev& BP = a.ball;
    const bool ball_lane = BALL && lb == NB;  // the first idle lane of the env carries the ball
    if (ball_lane) {
        // (JOBS: state and aerodynamic force - held over a simulate() call - come from the job of the substep before: chunks 50 .. 53)
        if constexpr (JOBS && V2P_LL_FAST_HANDOVER == 0) if (handed) {
            f4 c0, c1, c2, c3;
            cload4x2(hand + 4 * 50, hand + 4 * 51, c0, c1);
            cload4x2(hand + 4 * 52, hand + 4 * 53, c2, c3);
            bl[0] = c0.x; bl[1] = c0.y; bl[2] = c0.z; bl[3] = c0.w; bl[4] = c1.x; bl[5] = c1.y; bl[6] = c1.z; bl[7] = c1.w;
            bl[8] = c2.x; bl[9] = c2.y; bl[10] = c2.z; bl[11] = c2.w; bl[12] = c3.x; bl[13] = c3.y; bl[14] = c3.z; bl[15] = c3.w;
            bgot = true;
        }
        if (!bgot) {
#pragma unroll
            for (int k = 0; k < 13; ++k) bl[k] = BP.state[e * 13 + k];
#pragma unroll
            for (int k = 13; k < 16; ++k) bl[k] = 0.f;
        }
#pragma unroll
        for (int k = 16; k < BL_SLOTS; ++k) bl[k] = 0.f;
    }
    long long tprev = DIAG && a.prof ? clock64() : 0;
    const long long wt0 = DIAG && a.wave_times ? wall_clock64() : 0;
    const int key_pred = DIAG && a.wave_times ? a.pair_key[e] : 0;  // what the launch order was built from (diagnostics)
    int tsum = 0, tmaxs = 0;
    V3 r{0.f, 0.f, 0.f};
    int ksum = 0, kdep = 0;  // contact load of this env over the launch (pairing key)
#if defined(V2P_LL_TIMELINE)
    // build-time diagnostics (tools/mkvariant.sh): wall-clock start / end of every JOB of the production kernels, by workgroup index
    const long long tl0 = a.wave_times ? wall_clock64() : 0, tc0 = a.wave_times ? clock64() : 0;
    auto timeline = [&]() {
        if (!DIAG && a.wave_times && lane == 0) {
            long long* tl = a.wave_times + (int64_t)jid * 4;
            tl[0] = tl0; tl[1] = wall_clock64(); tl[2] = (long long)bid + ((long long)sjob << 24) + ((long long)(mono ? 1 : 0) << 28) + ((long long)ksum << 32);
            tl[3] = clock64() - tc0;  // shader cycles of the job: with the wall clock, the clock the job ran at
        }
    };
#else
    auto timeline = [&]() {};
#endif

    for (int sub = sub0; sub < sub1; ++sub) {
        // (the substep length behind an opaque scalar move: the dozen uniform expressions of it - 1 / h, h / n_iter, erp / hs, 1 / (1 + h damping),
        // margin / h ... - are then evaluated where they are used, a few scalar-operand instructions each, instead of being hoisted in front
        // of the substep loop and kept - spilled: in the racket + ball kernels - across all of it)
        float h = h_launch;
        if constexpr (OPAQUE_H) asm volatile("" : "+s"(h));
        const bool wrench_on = sub < P.hold_sub;
        const bool last = sub == nsub - 1;
        // a substep that is being recomputed by a job that gave up waiting for its predecessor: it publishes nothing (the predecessor does)
        const bool replay = JOBS && !mono && sub < jstart(sjob);
        LLPH(0);
        // per-link model constants are (re)loaded where they are used (L1/K$ hits) instead of pinning ~20 registers for the whole
        // kernel; the opaque index keeps the compiler from hoisting the loads back out of the substep loop
        int bo = b;
        asm volatile("" : "+v"(bo));
        const V3 lpos{S->local_pos[bo][0], S->local_pos[bo][1], S->local_pos[bo][2]};
        // ================================================================ pass 1: kinematics, root -> leaves by level
        V3 zw{0.f, 0.f, 0.f}, zv{0.f, 0.f, 0.f};
#if V2P_LL_KIN_JUMP
        {
            V3 wrel{0.f, 0.f, 0.f};
            kin_jump(q, x, w, xd, jq, wt, lpos, r, wrel);
            if (valid && dep >= 1) {
                const V3 pw = w - wrel, wpr = cross(pw, r);
                zw = cross(pw, wrel);
                zv = cross(pw, wpr);
            }
        }
#else
        for (int d = 1; d <= maxd; ++d) {
            const bool nc = (nonchain >> d) & 1;
            Q4 pq = pp(q, nc);
            V3 px = pp(x, nc), pw = pp(w, nc), pxd = pp(xd, nc);
            if (dep == d) {
                q = qnormalize(qmul(pq, jq));
                r = mul(q2mat(pq), lpos);
                x = px + r;
                V3 wrel = mul(q2mat(q), wt);
                w = pw + wrel;
                V3 wpr = cross(pw, r);
                xd = pxd + wpr;
                zw = cross(pw, wrel);
                zv = cross(pw, wpr);
            }
        }
#endif
        // (racket + ball: the ball lane and the ball x hull narrow phase run HERE, right after the kinematics, where a lane holds little
        // more than its pose and velocity: further down, next to the link's inertia blocks, their temporaries did not fit)
        if (BALL) {
            // pose and (start-of-substep) velocity of the racket's link, handed to the ball lane of the same env
            const int src = base + BP.racket_link;
            const Q4 wq = pull(q, src);
            const V3 wx = pull(x, src), ww = pull(w, src), wxd = pull(xd, src);
            if (ball_lane) {
                const V3 bp{bl[BL_POS], bl[BL_POS + 1], bl[BL_POS + 2]}, bv{bl[BL_VEL], bl[BL_VEL + 1], bl[BL_VEL + 2]}, bw{bl[BL_ANG], bl[BL_ANG + 1], bl[BL_ANG + 2]};
                bl[BL_V0] = bv.x; bl[BL_V0 + 1] = bv.y; bl[BL_V0 + 2] = bv.z;  // (the hull points' activation test reads the velocity of the start of the substep)
#pragma unroll
                for (int j = 2; j < NBREC; ++j) {
                    lds_vfloat* rk = bl + BL_RK + 16 * j;
                    rk[RK_A] = 0.f; rk[RK_LINK] = -1.f;
                    rk[RK_LAM] = 0.f; rk[RK_LAM + 1] = 0.f; rk[RK_LAM + 2] = 0.f;
                }
                if (sub % BP.sub_per_sim == 0) {
                    // the reference's bounce test on the ball height at the start of every simulate() call (apply_external_force_to_ball, :731-737)
                    // (system-scope accesses: with substep jobs the calls of one step run in different workgroups)
                    const int64_t e = env_here();
                    if (BP.has_bounce && live_env && !replay) {
                        if (sub == 0) flag_st(&BP.has_bounce_now[e], 0);
                        if (bp.z <= BP.bounce_height && !flag_ld(&BP.has_bounce[e])) {
                            flag_st(&BP.has_bounce[e], 1);
                            flag_st(&BP.has_bounce_now[e], 1);
                            BP.bounce_pos[e * 3] = bp.x; BP.bounce_pos[e * 3 + 1] = bp.y; BP.bounce_pos[e * 3 + 2] = bp.z;
                        }
                    }
                    // aerodynamic force, re-evaluated before every simulate() call (humanoid_smpl_im_mvae.py:711-739, utils/tennis_ball.py)
                    const float kf = 1.21f * 3.14159265358979f * 0.032f * 0.032f * 0.5f, cd = 0.55f;
                    const float sp = PHYS_SQRT(dot(bv, bv)), vs = sp == 0.f ? 1.f : sp;
                    const V3 vn = PHYS_RCP(vs) * bv;
                    const V3 vt = cross(vn, V3{0.f, 0.f, -1.f}), lt = cross(vt, vn);
                    const float vspin = PHYS_SQRT(dot(bw, bw)) * (1.f / 6.28318530717959f);
                    float cl = PHYS_RCP(2.f + fabsf(vs * PHYS_RCP(vspin * BP.spin_scale + 1e-6f)));
                    cl = vspin > 0.f ? -cl : cl;
                    const V3 F = (-kf * cd * vs) * bv - (kf * cl * vs * vs) * lt;
                    bl[BL_F] = F.x; bl[BL_F + 1] = F.y; bl[BL_F + 2] = F.z;
                }
                const V3 F{bl[BL_F], bl[BL_F + 1], bl[BL_F + 2]};
                const V3 bvs = bv + h * (V3{0.f, 0.f, P.gravity_z} + BP.inv_mass * F);  // free flight: v*
                bl[BL_VEL] = bvs.x; bl[BL_VEL + 1] = bvs.y; bl[BL_VEL + 2] = bvs.z;
                const float ih = PHYS_RCP(h), coff = P.contact_offset;
                // ---- ball x ground (speculative margin: the distance the ball can close within this substep)
                {
                    const float gap = bp.z - BP.radius;
                    const bool on = CONTACT && gap < coff + h * fmaxf(0.f, -bv.z);
                    float bias = gap >= 0.f ? gap * ih : fmaxf(P.erp * gap * ih, -P.max_depen);
                    const float rest = (bvs.z < -BP.bounce_thr && gap * ih + bvs.z < 0.f) ? BP.rest_ground * bvs.z : 3.0e38f;  // restitution
                    bias = fminf(bias, rest);
                    bl[BL_GA] = on ? 1.f : 0.f; bl[BL_GGAP] = gap; bl[BL_GBIAS] = bias;
                    if (TGS) bl[BL_GREST] = rest;
                    bl[BL_GLAM] = 0.f; bl[BL_GLAM + 1] = 0.f; bl[BL_GLAM + 2] = 0.f;
                }
                // ---- ball x the racket's solid cylinders: closest point, normal from the cylinder to the ball
                const M3 Rw = q2mat(wq);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    lds_vfloat* rk = bl + BL_RK + 16 * j;
                    bool on = false;
                    if (CONTACT && j < BP.ncyl) {
                        const V3 cw = wx + mul(Rw, V3{BP.cyl[j][0], BP.cyl[j][1], BP.cyl[j][2]}), aw = mul(Rw, V3{BP.cyl[j][3], BP.cyl[j][4], BP.cyl[j][5]});
                        const float hl = BP.cyl[j][6], rc = BP.cyl[j][7];
                        const V3 d = bp - cw;
                        const float t = dot(d, aw);
                        const V3 qv = d - t * aw;
                        const float rho = PHYS_SQRT(dot(qv, qv));
                        const float tc = fminf(fmaxf(t, -hl), hl), kq = rho > rc ? rc * PHYS_RCP(rho) : 1.f;
                        const V3 pt = cw + tc * aw + kq * qv;
                        const V3 ev = bp - pt;
                        const float dist = PHYS_SQRT(dot(ev, ev));
                        const V3 n = dist > 1e-9f ? PHYS_RCP(dist) * ev : (t >= 0.f ? aw : -aw);
                        const V3 rl = pt - wx;
                        const float vrel = dot(bv - wxd - cross(ww, rl), n);
                        const float gap = dist - BP.radius;
                        on = gap < coff + h * fmaxf(0.f, -vrel);
                        rk[RK_GAP] = gap;
                        rk[RK_RL] = rl.x; rk[RK_RL + 1] = rl.y; rk[RK_RL + 2] = rl.z;
                        rk[RK_N] = n.x; rk[RK_N + 1] = n.y; rk[RK_N + 2] = n.z;
                    }
                    rk[RK_A] = on ? 1.f : 0.f;
                    rk[RK_LAM] = 0.f; rk[RK_LAM + 1] = 0.f; rk[RK_LAM + 2] = 0.f;
                }
            }
            if (CONTACT && BP.body_contacts) {
                // ---- ball x the hulls of the links (every shape of the humanoid collides with the ball actor): each link lane tests its
                // own hull - bounding box first -, the nearest hull that the ball can reach within this substep carries the point
                const V3 bp{bl[BL_POS], bl[BL_POS + 1], bl[BL_POS + 2]}, bv0{bl[BL_V0], bl[BL_V0 + 1], bl[BL_V0 + 2]};
                const float coff = P.contact_offset;
                bool near = false;
                V3 cb{0.f, 0.f, 0.f};
                const M3 R = q2mat(q);
                if (valid && lb != BP.racket_link) {
                    cb = mulT(R, bp - x);  // ball centre, body frame
                    const V3 ac{S->aabb_c[bo][0], S->aabb_c[bo][1], S->aabb_c[bo][2]}, ae{S->aabb_e[bo][0], S->aabb_e[bo][1], S->aabb_e[bo][2]};
                    const V3 ex{fmaxf(fabsf(cb.x - ac.x) - ae.x, 0.f), fmaxf(fabsf(cb.y - ac.y) - ae.y, 0.f), fmaxf(fabsf(cb.z - ac.z) - ae.z, 0.f)};
                    const V3 dv = bv0 - xd;
                    const float vmax = PHYS_SQRT(dot(dv, dv)) + PHYS_SQRT(dot(w, w)) * (PHYS_SQRT(dot(ac, ac)) + PHYS_SQRT(dot(ae, ae)));
                    const float reach = BP.radius + coff + h * vmax;
                    near = dot(ex, ex) < reach * reach && S->hull_count[bo] > 0;
                }
                if (any64(near)) {
                    float gap = 3.0e38f;
                    V3 rlw{0.f, 0.f, 0.f}, nw{0.f, 0.f, 1.f};
                    if (near) {
                        const int hv0 = S->hull_offsets[bo], hnv = S->hull_count[bo];
                        const f4 hc = ball_hull_distance(S, hv0, hnv, cb);
                        V3 pb{hc.x, hc.y, hc.z};
                        float dist = hc.w;
                        V3 nb;
                        if (dist > 1e-6f) nb = PHYS_RCP(dist) * (cb - pb);
                        else {  // centre inside the hull: out along the direction from the centre of the bounding box
                            const V3 ev = cb - V3{S->aabb_c[bo][0], S->aabb_c[bo][1], S->aabb_c[bo][2]};
                            const float l = PHYS_SQRT(dot(ev, ev));
                            nb = l > 1e-9f ? PHYS_RCP(l) * ev : V3{0.f, 0.f, 1.f};
                            dist = 0.f;
                            pb = cb;
                        }
                        nw = mul(R, nb);
                        rlw = mul(R, pb);
                        const float vrel = dot(bv0 - xd - cross(w, rlw), nw);
                        const float g = dist - BP.radius;
                        if (g < coff + h * fmaxf(0.f, -vrel)) gap = g;
                    }
                    // the (up to) three nearest of the env's candidates, nearest first (ties: the lower link): one point per overlapping link,
                    // as PhysX generates one per overlapping pair - the LDS block holds three next to the cylinders' two
#pragma unroll 1
                    for (int k = 0; k < NBREC - 2; ++k) {
                        float gmin = gap;
#pragma unroll
                        for (int sh = 1; sh < 32; sh <<= 1) gmin = fminf(gmin, __shfl_xor(gmin, sh));
                        const unsigned long long wb = __ballot(gap == gmin && gap < 3.0e38f);
                        if (!wb) break;
                        const unsigned wmine = half ? (unsigned)(wb >> 32) : (unsigned)wb;
                        if (wmine && lb == __ffs(wmine) - 1) {
                            lds_vfloat* rk = bl + BL_RK + 16 * (2 + k);
                            rk[RK_A] = 1.f;
                            rk[RK_GAP] = gap;
                            rk[RK_RL] = rlw.x; rk[RK_RL + 1] = rlw.y; rk[RK_RL + 2] = rlw.z;
                            rk[RK_N] = nw.x; rk[RK_N + 1] = nw.y; rk[RK_N + 2] = nw.z;
                            rk[RK_LINK] = (float)lb;
                            gap = 3.0e38f;  // taken
                        }
                    }
                }
            }
        }
        // ---- per link, all lanes at once: joint torque, body inertia at its origin (world axes), bias force
        // mass properties are needed once per substep: reload them (L1/K$ hits) instead of pinning 10 registers for the
        // whole kernel; the opaque index keeps the loads from being hoisted back out of the loop
        const float mass = S->mass[bo];
        const V3 com{S->com[bo][0], S->com[bo][1], S->com[bo][2]};
        const Sym3 Ib{S->inertia[bo][0], S->inertia[bo][1], S->inertia[bo][2], S->inertia[bo][3], S->inertia[bo][4], S->inertia[bo][5]};
        V3 tau{0.f, 0.f, 0.f};
        float lsgn[3] = {0.f, 0.f, 0.f}, lbias[3] = {0.f, 0.f, 0.f}, llam[3] = {0.f, 0.f, 0.f};  // LIMITS: this joint's rows (sign 0 = none)
        bool limact = false;
        Sym3 Kd{1.f, 0.f, 0.f, 1.f, 0.f, 1.f};  // LIMITS: joint-space inverse inertia of this joint, body axes
        Sym3 A;
        M3 B;
        Sym3 C{mass, 0.f, 0.f, mass, 0.f, mass};
        V3 pn, pf;
        {
            const M3 R = q2mat(q);
            const float kp = S->kp[bo], kd = S->kd[bo];
            const V3 qe = b != 0 ? quat_to_expmap_stable(jq) : V3{0.f, 0.f, 0.f};
            if (b != 0) tau = mul(R, kp * (park_get3(PARK_TAR) - qe) - (kd + h * kp) * wt);
            if (LIMITS) {
                limact = false;
                const float ih = PHYS_RCP(h);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float lo = S->limit_lo[bo][i], hi = S->limit_hi[bo][i], qi = i == 0 ? qe.x : (i == 1 ? qe.y : qe.z);
                    const bool on = P.joint_limits && valid && b != 0 && hi - lo < 6.28f;
                    const float clo = qi - lo, chi = hi - qi;
                    const float gap = clo <= chi ? clo : chi;
                    lsgn[i] = on ? (clo <= chi ? 1.f : -1.f) : 0.f;
                    lbias[i] = TGS ? gap : (gap >= 0.f ? gap * ih : fmaxf(P.erp * gap * ih, -P.max_depen));  // (TGS: the gap, advanced slice by slice)
                    llam[i] = 0.f;
                    limact = limact || on;
                }
            }
            V3 dc = mul(R, com);
            V3 k0 = mul(Ib, V3{R.m[0], R.m[1], R.m[2]});
            V3 k1 = mul(Ib, V3{R.m[3], R.m[4], R.m[5]});
            V3 k2 = mul(Ib, V3{R.m[6], R.m[7], R.m[8]});
            V3 r0 = row(R, 0), r1 = row(R, 1), r2 = row(R, 2);
            Sym3 Ic{dot(r0, k0), dot(r0, k1), dot(r0, k2), dot(r1, k1), dot(r1, k2), dot(r2, k2)};
            float dd = dot(dc, dc);
            A = Sym3{Ic.xx + mass * (dd - dc.x * dc.x), Ic.xy - mass * dc.x * dc.y, Ic.xz - mass * dc.x * dc.z,
                     Ic.yy + mass * (dd - dc.y * dc.y), Ic.yz - mass * dc.y * dc.z, Ic.zz + mass * (dd - dc.z * dc.z)};
            B.m[0] = 0.f;            B.m[1] = -mass * dc.z;  B.m[2] = mass * dc.y;
            B.m[3] = mass * dc.z;    B.m[4] = 0.f;           B.m[5] = -mass * dc.x;
            B.m[6] = -mass * dc.y;   B.m[7] = mass * dc.x;   B.m[8] = 0.f;
            V3 wwd = cross(w, cross(w, dc));
            pf = mass * (wwd - V3{0.f, 0.f, P.gravity_z});
            pn = cross(w, mul(Ic, w)) + cross(dc, pf);
            if (b == 0 && wrench_on) {
                V3 extF, extT;
                if (a.actions) {
                    extF = V3{wrench_park[PARK_TAR * 64], wrench_park[(PARK_TAR + 1) * 64], wrench_park[(PARK_TAR + 2) * 64]};
                    extT = V3{wrench_park[25 + PARK_TAR * 64], wrench_park[25 + (PARK_TAR + 1) * 64], wrench_park[25 + (PARK_TAR + 2) * 64]};
                } else {
                    const float* cw = a.ctrl + env_here() * CTRL_SLOTS;
                    extF = V3{ldin(&cw[CT_FORCE + 0]), ldin(&cw[CT_FORCE + 1]), ldin(&cw[CT_FORCE + 2])};
                    extT = V3{ldin(&cw[CT_TORQUE + 0]), ldin(&cw[CT_TORQUE + 1]), ldin(&cw[CT_TORQUE + 2])};
                }
                pn = pn - extT - cross(dc, extF);  // force acts at the root COM
                pf = pf - extF;
            }
        }

        // pose of the link: next used by contact generation, then by the integration
        park[PARK_Q * 64] = q.x; park[(PARK_Q + 1) * 64] = q.y; park[(PARK_Q + 2) * 64] = q.z; park[(PARK_Q + 3) * 64] = q.w;
        park_put3(PARK_X, x);
        LLPH(1);
        park_vel(w, xd);
        // ================================================================ pass 2: articulated inertia, leaves -> root by level
        Sym3 Di{1.f, 0.f, 0.f, 1.f, 0.f, 1.f};
        M3 E;
#pragma unroll
        for (int i = 0; i < 9; ++i) E.m[i] = 0.f;
        V3 u{0.f, 0.f, 0.f};
        for (int d = maxd; d >= 1; --d) {
            Sym3 cA{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cC{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            M3 cB;
#pragma unroll
            for (int i = 0; i < 9; ++i) cB.m[i] = 0.f;
            V3 cn{0.f, 0.f, 0.f}, cf{0.f, 0.f, 0.f};
            if (dep == d) {
                Sym3 D{A.xx + aug, A.xy, A.xz, A.yy + aug, A.yz, A.zz + aug};
                Di = inv(D);
                E = mul(Di, B);
                u = tau - pn;
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
                // shift to the parent origin: S = [r]x Ca (columns r x Ca_col)
                V3 s0 = cross(r, V3{Ca.xx, Ca.xy, Ca.xz}), s1 = cross(r, V3{Ca.xy, Ca.yy, Ca.yz}), s2 = cross(r, V3{Ca.xz, Ca.yz, Ca.zz});
                cB.m[0] = Ba.m[0] + s0.x; cB.m[1] = Ba.m[1] + s1.x; cB.m[2] = Ba.m[2] + s2.x;
                cB.m[3] = Ba.m[3] + s0.y; cB.m[4] = Ba.m[4] + s1.y; cB.m[5] = Ba.m[5] + s2.y;
                cB.m[6] = Ba.m[6] + s0.z; cB.m[7] = Ba.m[7] + s1.z; cB.m[8] = Ba.m[8] + s2.z;
                V3 t10 = cross(r, row(Ba, 0)), t11 = cross(r, row(Ba, 1)), t12 = cross(r, row(Ba, 2));
                V3 sr0{s0.x, s1.x, s2.x}, sr1{s0.y, s1.y, s2.y}, sr2{s0.z, s1.z, s2.z};
                V3 t20 = cross(r, sr0), t21 = cross(r, sr1), t22 = cross(r, sr2);
                cA = Sym3{Aa.xx + 2.f * t10.x + t20.x, Aa.xy + t10.y + t11.x + t20.y, Aa.xz + t10.z + t12.x + t20.z,
                          Aa.yy + 2.f * t11.y + t21.y, Aa.yz + t11.z + t12.y + t21.z, Aa.zz + 2.f * t12.z + t22.z};
                cC = Ca;
                cn = pan + cross(r, paf);
                cf = paf;
            }
            // parents (depth d-1) pull their children's contributions; contributions of lanes that are not at depth d are zero
            if ((nonchain >> d) & 1) {
                A = A + mask(has0, from_next(cA));
                C = C + mask(has0, from_next(cC));
                M3 t = from_next(cB);
#pragma unroll
                for (int i = 0; i < 9; ++i) B.m[i] += has0 ? t.m[i] : 0.f;
                pn = pn + mask(has0, from_next(cn));
                pf = pf + mask(has0, from_next(cf));
            } else {
                // every link of this level directly follows its parent: whatever the next lane contributes is this lane's child's
                // (lanes at other depths contribute zeros), so the shifted values are added unmasked (DPP operand of the add)
                A = A + from_next(cA);
                C = C + from_next(cC);
                M3 t = from_next(cB);
#pragma unroll
                for (int i = 0; i < 9; ++i) B.m[i] += t.m[i];
                pn = pn + from_next(cn);
                pf = pf + from_next(cf);
            }
            if ((multi >> d) & 1) {
#pragma unroll 1
                for (int sl = 1; sl <= 2; ++sl) {  // one extra child at a time keeps the register peak down
                    const int cl = sl == 1 ? cl1 : cl2;
                    const bool hs = sl == 1 ? has1 : has2;
                    A = A + mask(hs, pull(cA, cl));
                    C = C + mask(hs, pull(cC, cl));
                    M3 t1 = pull(cB, cl);
#pragma unroll
                    for (int i = 0; i < 9; ++i) B.m[i] += hs ? t1.m[i] : 0.f;
                    pn = pn + mask(hs, pull(cn, cl));
                    pf = pf + mask(hs, pull(cf, cl));
                }
            }
        }

        LLPH(2);
        unpark_vel(w, xd);
        // ================================================================ root: 6x6 solve (lane 0 of each env)
        Blocks Lam;  // operational-space inverse inertia of this lane's link (root: inverse articulated inertia)
        Lam.A = Lam.C = Sym3{1.f, 0.f, 0.f, 1.f, 0.f, 1.f};
#pragma unroll
        for (int i = 0; i < 9; ++i) Lam.B.m[i] = 0.f;
        V3 al{0.f, 0.f, 0.f}, ac{0.f, 0.f, 0.f}, dw{0.f, 0.f, 0.f}, dv{0.f, 0.f, 0.f};
        if (lb == 0) {
            float a6[21], inv6[21];
            a6[tri(0, 0)] = A.xx; a6[tri(1, 0)] = A.xy; a6[tri(2, 0)] = A.xz; a6[tri(1, 1)] = A.yy; a6[tri(2, 1)] = A.yz; a6[tri(2, 2)] = A.zz;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) a6[tri(3 + i, j)] = B.m[3 * j + i];
            a6[tri(3, 3)] = C.xx; a6[tri(4, 3)] = C.xy; a6[tri(5, 3)] = C.xz; a6[tri(4, 4)] = C.yy; a6[tri(5, 4)] = C.yz; a6[tri(5, 5)] = C.zz;
            spd6_inverse(a6, inv6);
            Lam.A = Sym3{inv6[tri(0, 0)], inv6[tri(1, 0)], inv6[tri(2, 0)], inv6[tri(1, 1)], inv6[tri(2, 1)], inv6[tri(2, 2)]};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) Lam.B.m[3 * i + j] = inv6[tri(3 + j, i)];
            Lam.C = Sym3{inv6[tri(3, 3)], inv6[tri(4, 3)], inv6[tri(5, 3)], inv6[tri(4, 4)], inv6[tri(5, 4)], inv6[tri(5, 5)]};
            al = -(mul(Lam.A, pn) + mul(Lam.B, pf));
            ac = -(V3{dot(col(Lam.B, 0), pn), dot(col(Lam.B, 1), pn), dot(col(Lam.B, 2), pn)} + mul(Lam.C, pf));
            dw = h * al;
            dv = h * ac;
            w = w + dw;
            xd = xd + dv;
        }

        // ================================================================ pass 3: accelerations -> v*, root -> leaves by level
        for (int d = 1; d <= maxd; ++d) {
            const bool nc = (nonchain >> d) & 1;
            V3 alp = pp(al, nc), acp = pp(ac, nc), dwp = pp(dw, nc), dvp = pp(dv, nc);
            if (dep == d) {
                V3 aw = alp + zw;
                V3 av = acp + cross(alp, r) + zv;
                V3 qdd = mul(Di, u + aug * aw) - aw - mul(E, av);
                al = aw + qdd;
                ac = av;
                dw = dwp + h * qdd;  // v* - v_old at the OLD configuration
                dv = dvp + cross(dwp, r);
                w = w + dw;
                xd = xd + dv;
            }
        }

        LLPH(3);
        if (CONTACT) park_vel(w, xd);
        float* const rootlam = park_all + (threadIdx.x >> 6) * LDS_FLOATS_PER_WAVE + PARK_SLOTS * 64 + 2 * BL_SLOTS + half * 24;
        if (PARK2 && CONTACT && lb == 0) {  // only the root's Lambda exists yet: 21 floats per env, out of the way while contacts are generated
            const float lv[21] = {Lam.A.xx, Lam.A.xy, Lam.A.xz, Lam.A.yy, Lam.A.yz, Lam.A.zz, Lam.B.m[0], Lam.B.m[1], Lam.B.m[2], Lam.B.m[3], Lam.B.m[4], Lam.B.m[5],
                                  Lam.B.m[6], Lam.B.m[7], Lam.B.m[8], Lam.C.xx, Lam.C.xy, Lam.C.xz, Lam.C.yy, Lam.C.yz, Lam.C.zz};
#pragma unroll
            for (int k = 0; k < 21; ++k) rootlam[k] = lv[k];
        }
        if (PARK2 && CONTACT) {  // (the other lanes' values are placeholders until the recursion writes them)
            Lam.A = Lam.C = Sym3{1.f, 0.f, 0.f, 1.f, 0.f, 1.f};
#pragma unroll
            for (int i = 0; i < 9; ++i) Lam.B.m[i] = 0.f;
        }
        q = Q4{park[PARK_Q * 64], park[(PARK_Q + 1) * 64], park[(PARK_Q + 2) * 64], park[(PARK_Q + 3) * 64]};
        x = park_get3(PARK_X);
        int cnt = 0;
        ContactStore<PARK3> CS(park);
#pragma unroll
        for (int c = 0; c < 4; ++c) { CS.set_cr(c, V3{0.f, 0.f, 0.f}); CS.set_bias(c, 0.f); CS.set_lam(c, V3{0.f, 0.f, 0.f}); }
        if (CONTACT) {
            // ============================================================ contact generation: every lane scans its own hull
            // pass A marks the candidate vertices (z < contact_offset) in a per-lane 64-bit mask; the manifold reduction then
            // walks only the candidates.  Hull vertices come from the LDS copy of the model (staged once per launch).
            const float coff = P.contact_offset;
            const int v0 = S->hull_offsets[bo], nv = S->hull_count[bo];
            // conservative culling: lowest point of the hull's body-frame bounding box (third row of the link's rotation)
            const float rz0 = 2.f * (q.x * q.z - q.w * q.y), rz1 = 2.f * (q.y * q.z + q.w * q.x), rz2 = 1.f - 2.f * (q.x * q.x + q.y * q.y);
            const float zlow = x.z + rz0 * S->aabb_c[bo][0] + rz1 * S->aabb_c[bo][1] + rz2 * S->aabb_c[bo][2] -
                               (fabsf(rz0) * S->aabb_e[bo][0] + fabsf(rz1) * S->aabb_e[bo][1] + fabsf(rz2) * S->aabb_e[bo][2]);
            const bool near = valid && (zlow < coff + 1e-4f);
            int pack = 0x0fffffff;  // four 7-bit vertex slots (127 = none), manifold size in bits 28..30
            long long tsub = DIAG && a.prof ? clock64() : 0;
            const unsigned long long nball = __ballot(near);
            LLSUB(16);
            if (nball) {
                // The hulls of the few links near the ground are scanned by GROUPS of 8 lanes, 8 groups per wave: in every round group g
                // takes the next near link of the WAVE (the near links of both envs in lane order), whichever env it belongs to - a
                // ragdoll with 17 near links that shares its wave with a standing humanoid needs 3 rounds instead of 5.  Lane gl of a
                // group takes vertices gl, gl+8, ... and the group combines with three DPP steps.  Every selection keeps the serial rule
                // "first index attaining the extreme": per lane indices ascend, across lanes ties go to the lower index; which group
                // scans a link changes nothing in its numbers.
                const unsigned nm = half ? (unsigned)(nball >> 32) : (unsigned)nball;
                const int nr0 = __popc((unsigned)nball), nr1 = __popc((unsigned)(nball >> 32));
                const int ntot = nr0 + nr1;
                const int rounds = (ntot + 7) >> 3;
                const int myk = (half ? nr0 : 0) + __popc(nm & ((1u << lb) - 1u));  // rank of this link among the near links of the wave
                const int grp = lane >> 3, gl = lane & 7;
                int* const scr = (int*)(park_all + (threadIdx.x >> 6) * LDS_FLOATS_PER_WAVE + PARK_SCR * 64);  // k-th near link -> its lane
                if (near) scr[myk] = lane;
                if (DIAG && a.prof && ((a.prof_heavy ? blockIdx.x < 8u : (blockIdx.x & 63) == 0)) && lane == 0) atomicAdd((unsigned long long*)&a.prof[13], (unsigned long long)rounds);
                for (int rd = 0; rd < rounds; ++rd) {
                    const int kk = 8 * rd + grp;
                    const bool gon = kk < ntot;
                    const int src = gon ? scr[kk] : lane;
                    // the group works in the frame of ITS link: pose pulled from the owner lane (7 values instead of 12)
                    const M3 RL = q2mat(pull(q, src));
                    const V3 xL = pull(x, src);
                    const float r6 = RL.m[6], r7 = RL.m[7], r8 = RL.m[8], xz = xL.z;
                    const int v0L = __builtin_amdgcn_ds_bpermute(src << 2, v0), nvL = __builtin_amdgcn_ds_bpermute(src << 2, nv);
                    // (per-env shapes: the link may belong to the other env of the wave)
                    ConstShape* const SL = MULTI ? (ConstShape*)(a.shapes + __builtin_amdgcn_ds_bpermute(src << 2, sid)) : S;
                    auto hullv = [&](int idx) -> float4 { return make_float4(SL->hull_verts[idx][0], SL->hull_verts[idx][1], SL->hull_verts[idx][2], 0.f); };
                    // ---- one pass over the hull: 8 vertices per lane (all loads in flight at once), ground-plane coordinates kept
                    // for the manifold reduction; candidates (z < contact_offset) as a bit mask, deepest one tracked
                    const float r0 = RL.m[0], r1 = RL.m[1], r2 = RL.m[2], xx = xL.x;
                    const float r3 = RL.m[3], r4 = RL.m[4], r5 = RL.m[5], xy = xL.y;
                    const unsigned gbit = 1u << gl;
                    float4 vv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int i = 8 * k + gl;
                        vv[k] = hullv(v0L + ((gon && i < nvL) ? i : 0));
                    }
                    float px[8], py[8];
                    unsigned clo = 0u, chi = 0u;
                    int k0 = -1;
                    float zmin = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int i = 8 * k + gl;
                        const bool in = gon && i < nvL;
                        const float z = xz + r6 * vv[k].x + r7 * vv[k].y + r8 * vv[k].z;
                        px[k] = xx + r0 * vv[k].x + r1 * vv[k].y + r2 * vv[k].z;
                        py[k] = xy + r3 * vv[k].x + r4 * vv[k].y + r5 * vv[k].z;
                        const bool c = in && (z < coff);
                        const unsigned bit = c ? gbit << (8 * (k & 3)) : 0u;
                        if (k < 4) clo |= bit; else chi |= bit;
                        const bool better = c && (k0 < 0 || z < zmin);
                        k0 = better ? i : k0;
                        zmin = better ? z : zmin;
                    }
                    clo |= grp_xor1(clo); chi |= grp_xor1(chi);
                    clo |= grp_xor2(clo); chi |= grp_xor2(chi);
                    clo |= grp_mirror(clo); chi |= grp_mirror(chi);
                    grp_argmin(zmin, k0);
                    const unsigned long long cm = ((unsigned long long)chi << 32) | clo;
                    const int cntg = __popc(clo) + __popc(chi);
                    unsigned long long t = cm;
                    int s0 = t ? __ffsll((long long)t) - 1 : -1; t &= t - 1;
                    int s1 = t ? __ffsll((long long)t) - 1 : -1; t &= t - 1;
                    int s2 = t ? __ffsll((long long)t) - 1 : -1; t &= t - 1;
                    int s3 = t ? __ffsll((long long)t) - 1 : -1;
                    int ns = cntg < 4 ? cntg : 4;
                    const bool big = gon && cntg > 4;
                    if (any64(big)) {
                        if (DIAG && a.prof && ((a.prof_heavy ? blockIdx.x < 8u : (blockIdx.x & 63) == 0)) && lane == 0) atomicAdd((unsigned long long*)&a.prof[15], 1ull);
                        // manifold reduction: deepest, farthest from it, extreme on either side of that line
                        const int k0s = k0 < 0 ? 0 : k0;
                        const float4 u0 = hullv(v0L + k0s);
                        const float p0x = xx + r0 * u0.x + r1 * u0.y + r2 * u0.z;
                        const float p0y = xy + r3 * u0.x + r4 * u0.y + r5 * u0.z;
                        const unsigned long long remB = big ? (cm & ~(1ull << k0s)) : 0ull;
                        const unsigned blo = (unsigned)remB, bhi = (unsigned)(remB >> 32);
                        int k1 = -1;
                        float best = -1.f;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const bool on = ((k < 4 ? blo : bhi) & (gbit << (8 * (k & 3)))) != 0u;
                            const float dx = px[k] - p0x, dy = py[k] - p0y;
                            const float d2 = dx * dx + dy * dy;
                            const bool take = on && (d2 > best);
                            best = take ? d2 : best;
                            k1 = take ? 8 * k + gl : k1;
                        }
                        grp_argmax(best, k1);
                        const int k1s = k1 < 0 ? 0 : k1;
                        const float4 u1 = hullv(v0L + k1s);
                        const float ex = xx + r0 * u1.x + r1 * u1.y + r2 * u1.z - p0x;
                        const float ey = xy + r3 * u1.x + r4 * u1.y + r5 * u1.z - p0y;
                        const unsigned long long remC = remB & ~(1ull << k1s);
                        const unsigned elo = (unsigned)remC, ehi = (unsigned)(remC >> 32);
                        int k2 = -1, k3 = -1;
                        float amax = 0.f, amin = 0.f;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const bool on = ((k < 4 ? elo : ehi) & (gbit << (8 * (k & 3)))) != 0u;
                            const float dx = px[k] - p0x, dy = py[k] - p0y;
                            const float area = ex * dy - ey * dx;
                            const bool up = on && area > amax;
                            const bool dn = on && area < amin;
                            amax = up ? area : amax; k2 = up ? 8 * k + gl : k2;
                            amin = dn ? area : amin; k3 = dn ? 8 * k + gl : k3;
                        }
                        grp_argmax(amax, k2);
                        grp_argmin(amin, k3);
                        if (big) {
                            s0 = k0; s1 = k1;
                            s2 = k2 >= 0 ? k2 : k3;
                            s3 = k2 >= 0 ? k3 : -1;
                            ns = 2 + (k2 >= 0 ? 1 : 0) + (k3 >= 0 ? 1 : 0);
                        }
                    }
                    const int gpack = (s0 & 0x7f) | ((s1 & 0x7f) << 7) | ((s2 & 0x7f) << 14) | ((s3 & 0x7f) << 21) | (ns << 28);
                    const int got = __builtin_amdgcn_ds_bpermute(((myk & 7) << 3) << 2, gpack);
                    pack = (near && (myk >> 3) == rd) ? got : pack;
                }
            }
            LLSUB(17);
            int sel4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int sv = (pack >> (7 * c)) & 0x7f;
                sel4[c] = sv == 0x7f ? -1 : sv;
            }
            cnt = (pack >> 28) & 7;
            if (any64(cnt > 0)) {
                const M3 R = q2mat(q);
                const float ih = 1.f / h;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 uu = hullv(v0 + (sel4[c] < 0 ? 0 : sel4[c]));
                    const V3 crc = mul(R, V3{uu.x, uu.y, uu.z});
                    CS.set_cr(c, crc);
                    float dz = x.z + crc.z - P.rest_offset;
                    CS.set_bias(c, TGS ? dz : (dz >= 0.f ? dz * ih : fmaxf(P.erp * dz * ih, -P.max_depen)));
                }
            }
            if (a.contact_ids && last && valid && live_env && !frozen) {  // diagnostics (v2p_sim_cfg.debug_contacts)
#pragma unroll
                for (int c = 0; c < 4; ++c) a.contact_ids[(env_here() * NB + b) * 4 + c] = c < cnt ? b * 64 + sel4[c] : -1;
            }
            if (a.contact_ids_sub && valid && live_env && !frozen) {  // diagnostics: the ids of every substep (parity tests)
#pragma unroll
                for (int c = 0; c < 4; ++c) a.contact_ids_sub[((env_here() * nsub + sub) * NB + b) * 4 + c] = c < cnt ? b * 64 + sel4[c] : -1;
            }

            LLSUB(18);
            if (PARK2 && lb == 0) {  // Lambda of the root back from its block
                float lv[21];
#pragma unroll
                for (int k = 0; k < 21; ++k) lv[k] = rootlam[k];
                Lam.A = Sym3{lv[0], lv[1], lv[2], lv[3], lv[4], lv[5]};
#pragma unroll
                for (int k = 0; k < 9; ++k) Lam.B.m[k] = lv[6 + k];
                Lam.C = Sym3{lv[15], lv[16], lv[17], lv[18], lv[19], lv[20]};
            }
            // (with a ball: the racket's link joins the touched links while the ball is in contact with a cylinder)
            // (point j of the ball block belongs to this lane's link: j = 0, 1 the racket's cylinders, j = 2 .. 4 the hull points)
            auto ball_rec_mine = [&](int j) -> bool { return j < 2 ? lb == BP.racket_link : lb == (int)bl[BL_RK + 16 * j + RK_LINK]; };
            bool myhull = false;  // this link's hull carries a ball point
            if (BALL) {
#pragma unroll
                for (int j = 2; j < NBREC; ++j) myhull = myhull || (bl[BL_RK + 16 * j + RK_A] != 0.f && lb == (int)bl[BL_RK + 16 * j + RK_LINK]);
            }
            const bool ballhit = BALL && valid && ((lb == BP.racket_link && (bl[BL_RK + RK_A] != 0.f || bl[BL_RK + 16 + RK_A] != 0.f)) || myhull);
            const bool ballground = BALL && ball_lane && bl[BL_GA] != 0.f;
            const unsigned long long tb = __ballot(valid && (cnt > 0 || ballhit));
            const unsigned m0 = (unsigned)tb, m1 = (unsigned)(tb >> 32);
            if (LIMITS && any64(limact)) {
                // speculative activation of the limit rows (the model is stated in oracle/phys/v2p_phys_oracle.c): a row exists in this substep
                // only while its DOF is within limit_margin of the limit or would reach it at the approach rate of v* (the joint rate after
                // the unconstrained update: link velocity minus the parent's, body axes).  Joints far from their limits are no stops of the
                // walk and Lambda is not carried down to them.
                const V3 wv = PARK2 ? park_get3(PARK_W0) : w;
                const V3 pwv = pp(wv, true);
                const V3 om = mulT(q2mat(q), wv - pwv);
                const float mrate = P.limit_margin * PHYS_RCP(h);
                const float lscale = TGS ? PHYS_RCP(h) : 1.f;  // (TGS keeps the gap itself: gap / h is what the test compares, negative when violated either way)
                bool anyrow = false;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float omi = i == 0 ? om.x : (i == 1 ? om.y : om.z);
                    const bool on = lsgn[i] != 0.f && lbias[i] * lscale < mrate + fmaxf(0.f, -lsgn[i] * omi);  // (violated: lbias < 0)
                    if (!on) lsgn[i] = 0.f;
                    anyrow = anyrow || on;
                }
                limact = anyrow;
            }
            const unsigned long long lmb = (LIMITS && !(V2P_LL_EXP & 16)) ? __ballot(valid && limact) : 0ull;
            const unsigned lm0 = (unsigned)lmb, lm1 = (unsigned)(lmb >> 32);
            if (DIAG && a.wave_times) { const int tt = __popc(half ? m1 : m0); tsum += tt; tmaxs = tt > tmaxs ? tt : tmaxs; }
            if (last) {
                const unsigned mine = half ? m1 : m0;
                ksum = __popc(mine);
                kdep = 0;
                for (unsigned t = mine; t; t &= t - 1) { const int dd = M.depth[__ffs(t) - 1]; kdep = dd > kdep ? dd : kdep; }
            }
            if (DIAG && a.prof && ((a.prof_heavy ? blockIdx.x < 8u : (blockIdx.x & 63) == 0)) && lane == 0) { atomicAdd((unsigned long long*)&a.prof[9], (unsigned long long)(__popc(m0) + __popc(m1))); atomicAdd((unsigned long long*)&a.prof[10], 1ull); }
            const bool sweep_on = ((m0 | m1 | ((V2P_LL_EXP & 32) ? 0u : (lm0 | lm1))) || (BALL && any64(ballground))) && P.n_iter > 0;
            if (PARK2 && !sweep_on) unpark_vel(w, xd);
            if (sweep_on) {
                // deepest touched link of either env: links below it are never read during the sweep, so Lambda and the
                // per-update propagation stop there; their velocities catch up once at the end (the propagation is linear)
                // (each env stops at ITS deepest touched link, so its arithmetic does not depend on which env shares the wave)
                int dn0 = 0, dn1 = 0;
                for (unsigned t = (V2P_LL_EXP & 8) ? m0 : (m0 | lm0); t; t &= t - 1) { const int dd = M.depth[__ffs(t) - 1]; dn0 = dd > dn0 ? dd : dn0; }
                for (unsigned t = (V2P_LL_EXP & 8) ? m1 : (m1 | lm1); t; t &= t - 1) { const int dd = M.depth[__ffs(t) - 1]; dn1 = dd > dn1 ? dd : dn1; }
                const int dneed = dn0 > dn1 ? dn0 : dn1, dmin = dn0 < dn1 ? dn0 : dn1;
                const bool insweep = dep <= (half ? dn1 : dn0);  // this link moves with every update; the others catch up afterwards
                LLPH(4);
                // ======================================================== Lambda_b, root -> leaves by level
                for (int d = 1; d <= dneed; ++d) {
                    Blocks Lp;
                    const bool nc = (nonchain >> d) & 1;
                    Lp.A = pp(Lam.A, nc);
                    Lp.B = pp(Lam.B, nc);
                    Lp.C = pp(Lam.C, nc);
                    if (dep == d) {
                        // G = X Lp X^T: Ga = La ; Gb = La [r]x + Lb ; Gc = Lc - [r]x Lb + Gb^T [r]x
                        V3 la0{Lp.A.xx, Lp.A.xy, Lp.A.xz}, la1{Lp.A.xy, Lp.A.yy, Lp.A.yz}, la2{Lp.A.xz, Lp.A.yz, Lp.A.zz};
                        M3 Gb;
                        {
                            V3 g0 = cross(la0, r) + row(Lp.B, 0), g1 = cross(la1, r) + row(Lp.B, 1), g2 = cross(la2, r) + row(Lp.B, 2);
                            Gb.m[0] = g0.x; Gb.m[1] = g0.y; Gb.m[2] = g0.z; Gb.m[3] = g1.x; Gb.m[4] = g1.y; Gb.m[5] = g1.z; Gb.m[6] = g2.x; Gb.m[7] = g2.y; Gb.m[8] = g2.z;
                        }
                        Sym3 Gc;
                        {
                            V3 q0 = cross(col(Gb, 0), r), q1 = cross(col(Gb, 1), r), q2 = cross(col(Gb, 2), r);
                            V3 n0 = cross(r, col(Lp.B, 0)), n1 = cross(r, col(Lp.B, 1)), n2 = cross(r, col(Lp.B, 2));
                            Gc.xx = Lp.C.xx + q0.x - n0.x;
                            Gc.xy = Lp.C.xy + q0.y - n1.x;
                            Gc.xz = Lp.C.xz + q0.z - n2.x;
                            Gc.yy = Lp.C.yy + q1.y - n1.y;
                            Gc.yz = Lp.C.yz + q1.z - n2.y;
                            Gc.zz = Lp.C.zz + q2.z - n2.z;
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
                        M3 t;
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 3; ++j) t.m[3 * i + j] = aug * dot(row(DiM, i), col(H1, j)) - dot(row(E, i), col(H2, j));
                        Lam.A = Sym3{Di.xx + t.m[0], Di.xy + 0.5f * (t.m[1] + t.m[3]), Di.xz + 0.5f * (t.m[2] + t.m[6]), Di.yy + t.m[4],
                                     Di.yz + 0.5f * (t.m[5] + t.m[7]), Di.zz + t.m[8]};
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                V3 gcj = j == 0 ? V3{Gc.xx, Gc.xy, Gc.xz} : (j == 1 ? V3{Gc.xy, Gc.yy, Gc.yz} : V3{Gc.xz, Gc.yz, Gc.zz});
                                Lam.B.m[3 * i + j] = aug * dot(row(DiM, i), col(Gb, j)) - dot(row(E, i), gcj);
                            }
                        Lam.C = Gc;
                        if (LIMITS && limact) {
                            const Sym3 K{Lam.A.xx - 2.f * H1.m[0] + Lp.A.xx, Lam.A.xy - H1.m[1] - H1.m[3] + Lp.A.xy, Lam.A.xz - H1.m[2] - H1.m[6] + Lp.A.xz,
                                         Lam.A.yy - 2.f * H1.m[4] + Lp.A.yy, Lam.A.yz - H1.m[5] - H1.m[7] + Lp.A.yz, Lam.A.zz - 2.f * H1.m[8] + Lp.A.zz};
                            const M3 R = q2mat(Q4{park[PARK_Q * 64], park[(PARK_Q + 1) * 64], park[(PARK_Q + 2) * 64], park[(PARK_Q + 3) * 64]});
                            const V3 c0 = col(R, 0), c1 = col(R, 1), c2 = col(R, 2);
                            const V3 k0 = mul(K, c0), k1 = mul(K, c1), k2 = mul(K, c2);
                            Kd = Sym3{dot(c0, k0), dot(c0, k1), dot(c0, k2), dot(c1, k1), dot(c1, k2), dot(c2, k2)};
                        }
                    }
                }

                // direction a = 0, 1, 2 of a ball x racket point: its normal and the tangent basis t1 = normalize(n x z) (n x x when n is
                // within 1e-3 of +-z), t2 = n x t1 - the oracle's rule
                auto ball_dirs = [&](V3 n, V3& t1, V3& t2) {
                    t1 = cross(n, V3{0.f, 0.f, 1.f});
                    if (dot(t1, t1) < 1e-6f) t1 = cross(n, V3{1.f, 0.f, 0.f});
                    t1 = rsqrtf(dot(t1, t1)) * t1;
                    t2 = cross(n, t1);
                };
                if (PARK2) unpark_vel(w, xd);
                if (BALL && ballhit) {
                    // row biases of the ball x racket points from the velocities before the sweep (restitution: Newton, against the approach speed)
                    const V3 bv{bl[BL_VEL], bl[BL_VEL + 1], bl[BL_VEL + 2]}, bw{bl[BL_ANG], bl[BL_ANG + 1], bl[BL_ANG + 2]};
                    const float ih = PHYS_RCP(h);
#pragma unroll
                    for (int j = 0; j < NBREC; ++j) {
                        lds_vfloat* rk = bl + BL_RK + 16 * j;
                        if (rk[RK_A] != 0.f && ball_rec_mine(j)) {
                            const V3 n{rk[RK_N], rk[RK_N + 1], rk[RK_N + 2]}, rl{rk[RK_RL], rk[RK_RL + 1], rk[RK_RL + 2]};
                            const float gap = rk[RK_GAP];
                            const float vn0 = dot(bv + cross(bw, -BP.radius * n) - xd - cross(w, rl), n);
                            float bias = gap >= 0.f ? gap * ih : fmaxf(P.erp * gap * ih, -P.max_depen);
                            const float rest = (vn0 < -BP.bounce_thr && gap * ih + vn0 < 0.f) ? (j < 2 ? BP.rest_racket : BP.rest_body) * vn0 : 3.0e38f;
                            bias = fminf(bias, rest);
                            rk[RK_BIAS] = bias;
                            if (TGS) rk[RK_REST] = rest;
                        }
                    }
                }
                LLPH(5);
                if (!PARK2) {
                    park_put3(PARK_W0, w);  // the sweep's total delta-velocity of a link = its velocity at the end - these
                    park_put3(PARK_XD0, xd);
                }
                // touched links whose parent is the touched link right before them (ascending): they continue a group (below)
                unsigned chain0 = 0u, chain1 = 0u;
                {
                    int prev = -1;
                    for (unsigned t = m0; t; t &= t - 1) {
                        const int nn = __ffs(t) - 1;
                        if ((int)((a.par_pack[nn / 12] >> (5 * (nn % 12))) & 31ull) == prev) chain0 |= 1u << nn;
                        prev = nn;
                    }
                    prev = -1;
                    for (unsigned t = m1; t; t &= t - 1) {
                        const int nn = __ffs(t) - 1;
                        if ((int)((a.par_pack[nn / 12] >> (5 * (nn % 12))) & 31ull) == prev) chain1 |= 1u << nn;
                        prev = nn;
                    }
                    chain0 &= ~lm0;  // the limit rows of a joint come between its parent's block and its own: no chaining into it
                    chain1 &= ~lm1;
                }
                // ======================================================== block Gauss-Seidel: k-th touched body of each env at once
                const float hs = h / (float)P.n_iter;  // TGS: length of a time slice
                float tgs_irem = 1.f / h;                // TGS: 1 / (time left in the substep) for separated points
                const float tgs_pen = P.erp / hs;
                auto rowbias = [&](float v) -> float { return TGS ? (v >= 0.f ? v * tgs_irem : fmaxf(tgs_pen * v, -P.max_depen)) : v; };
                static_assert(!VFRIC || (WALK && CONTACT), "the velocity-aligned friction frame exists in the walk form of the sweep");
                if constexpr (WALK) {
                // ---- the sweep as ONE WALK over the tree.  Solving the touched links one by one in ascending order visits them in depth-first
                // order, cyclically, iteration after iteration (back and forth with the experimental ALT switch below); between two of them only
                // the links on the tree path cur -> LCA -> next need anything:
                //   up    cur .. LCA: every link hands what its subtree has collected since it last did so (un_new, uf_new) to its parent,
                //         and the LCA answers what arrives with its own Lambda (Lambda_cc is the response of the whole system at c);
                //   down  LCA .. next: velocity change of a link = its parent's, carried over the joint, + the joint's answer to everything
                //         its subtree has collected so far (un_tot) - the relation the root -> leaves pass uses, valid here because a
                //         depth-first walk enters a subtree only after everything applied inside it has been handed up through its root.
                // (Dw, Dv) of a link = its velocity change since the start of the sweep, valid whenever the walk stands on it; w, xd keep
                // the velocities of the start of the sweep.  After the last iteration the walk returns to the root and ONE root -> leaves
                // pass moves every link.  By linearity the row updates see exactly the velocities of the one-by-one sweep with a
                // leaf -> root -> leaves propagation after every link; that costs (touched groups x 2 x depth) level steps per iteration,
                // the walk 2 x (edges of the subtree the touched links span): the same for two feet on the ground, about half for a
                // fallen humanoid with 7 scattered touched links - the critical path of a launch.
                // TGS advances the gaps with the velocities after every iteration: there the walk is closed after every iteration.
                V3 un_tot{0.f, 0.f, 0.f}, un_new{0.f, 0.f, 0.f}, uf_new{0.f, 0.f, 0.f}, Dw{0.f, 0.f, 0.f}, Dv{0.f, 0.f, 0.f};
                int cur0 = 0, cur1 = 0;            // link the walk of each env stands on
                bool live0 = false, live1 = false;  // the env has applied an impulse (until then all its changes are zero and its walk rests)
                // the move INTO each touched link, from the touched link before it (cyclically): depth of their lowest common ancestor |
                // depth of the link before << 4 | own depth << 8 | side-entry levels << 12 (12 bits).  (A link has depth + 1 ancestors-or-self: depths from ballots.)
                // LIMITS: the walk also stops at the joints that carry limit rows (their block comes right before the contact block of the link)
                const unsigned v0 = (LIMITS && !(V2P_LL_EXP & 4)) ? m0 | lm0 : m0, v1 = (LIMITS && !(V2P_LL_EXP & 4)) ? m1 | lm1 : m1;
                V3 jt_new{0.f, 0.f, 0.f};  // LIMITS: limit impulse of this joint not yet handed up (its reaction, -jt, goes to the parent)
                // ALT (experiment, off by default: V2P_LL_ALT_SWEEP): odd PGS sweeps visit the stops in DESCENDING order - minfo_rev = the move into a
                // stop from the stop AFTER it (same fields; the highest stop has none: a backward sweep starts on it, where the forward sweep ended)
                constexpr bool ALT = V2P_LL_ALT_SWEEP != 0 && !TGS;
                int minfo = 0, minfo_rev = 0;
                {
                    int p0 = v0 ? 31 - __clz(v0) : 0, p1 = v1 ? 31 - __clz(v1) : 0;
                    bool first = true;
                    for (unsigned s0 = v0, s1 = v1; s0 | s1; s0 &= s0 - 1, s1 &= s1 - 1) {
                        const int b0 = s0 ? __ffs(s0) - 1 : p0, b1 = s1 ? __ffs(s1) - 1 : p1;
                        const int selp = half ? p1 : p0, selb = half ? b1 : b0;
                        const bool ap = valid && ((desc >> selp) & 1), ab = valid && ((desc >> selb) & 1);
                        const unsigned long long bp = __ballot(ap), bb = __ballot(ab), bc = __ballot(ap && ab);
                        const int du = __popc(half ? (unsigned)(bp >> 32) : (unsigned)bp) - 1, dn = __popc(half ? (unsigned)(bb >> 32) : (unsigned)bb) - 1,
                                  dl = __popc(half ? (unsigned)(bc >> 32) : (unsigned)bc) - 1;
                        // (levels of the way up where the path link is not its parent's first child: it hands over through a pull, see walk_to)
                        const int sd = (half ? M.side_depths[p1] : M.side_depths[p0]) & ~((2 << dl) - 1);
                        // (bit 28: the link itself is not its parent's first child - the one-joint bounce of a limit block needs it)
                        if (valid && lb == selb && ((half ? s1 : s0) != 0u)) minfo = dl | (du << 4) | (dn << 8) | (sd << 12) | ((LIMITS && !firstchild) ? 1 << 28 : 0);
                        if (ALT && !first) {
                            // the same pair walked the other way: from the stop b (deeper in the order) back into the stop before it
                            const int sdr = (half ? M.side_depths[b1] : M.side_depths[b0]) & ~((2 << dl) - 1);
                            if (valid && lb == selp && ((half ? s1 : s0) != 0u)) minfo_rev = dl | (dn << 4) | (du << 8) | (sdr << 12) | ((LIMITS && !firstchild) ? 1 << 28 : 0);
                        }
                        first = false;
                        p0 = b0;
                        p1 = b1;
                    }
                }
                // one move of both walks: env h goes from cur_h to its next link when mv_h (info_h = the move, see minfo), else it rests
                auto walk_to = [&](int nx0, int nx1, int info0, int info1, bool mv0, bool mv1) {
                    long long tsub = DIAG && a.prof ? clock64() : 0;
                    // a resting env: empty ranges that do not widen the loops (LCA depth 15, depths 0)
                    // (readfirstlane: the compiler must see wave-uniform loop bounds, or it runs the level loops with per-lane exits)
                    const int pk0 = __builtin_amdgcn_readfirstlane(mv0 ? info0 : 0x00f), pk1 = __builtin_amdgcn_readfirstlane(mv1 ? info1 : 0x00f);
                    const int dl0 = pk0 & 15, dn0 = (pk0 >> 8) & 15;
                    const int dl1 = pk1 & 15, dn1 = (pk1 >> 8) & 15;
                    const int selc = half ? cur1 : cur0, seln = half ? nx1 : nx0, mydl = half ? dl1 : dl0;
                    const bool onc = valid && ((desc >> selc) & 1), onn = valid && ((desc >> seln) & 1);  // ancestors (or self) of cur / of next
                    // LIMITS: a way up on which no link holds anything to hand over is not walked (the stops at joints whose limit rows
                    // do not act - most of them - leave nothing behind; the lowest common ancestor has been current since the walk came
                    // down through it): that env's way up counts as resting
                    int uk0 = pk0, uk1 = pk1;
                    if constexpr (LIMITS) {
                        const bool pend = onc && dep > mydl && (un_new.x != 0.f || un_new.y != 0.f || un_new.z != 0.f || uf_new.x != 0.f || uf_new.y != 0.f || uf_new.z != 0.f ||
                                                                jt_new.x != 0.f || jt_new.y != 0.f || jt_new.z != 0.f);
                        const unsigned long long pb = __ballot(pend);
                        uk0 = __builtin_amdgcn_readfirstlane((unsigned)pb ? pk0 : 0x00f);
                        uk1 = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32) ? pk1 : 0x00f);
                    }
                    const int du0 = (uk0 >> 4) & 15, du1 = (uk1 >> 4) & 15, ul0 = uk0 & 15, ul1 = uk1 & 15, myul = half ? ul1 : ul0;
                    // depth of this lane's link if it is on the way up (cur .. LCA + 1) / on the way down (LCA + 1 .. next), else -1
                    const int updep = (onc && dep > myul) ? dep : -1, dndep = (onn && dep > mydl) ? dep : -1;
                    const int turndep = (onc && dep == myul) ? dep : -2;  // the LCA itself: where this env's move turns
                    const int ulmin = ul0 < ul1 ? ul0 : ul1, dlmin = dl0 < dl1 ? dl0 : dl1;
                    // levels (bit d = the links at depth d hand over) that need the long form: a walk turns at their parent, or a path link
                    // is not the first child of its parent (it does not sit in the lane next to it)
                    const unsigned sideb = ((unsigned)(uk0 >> 12) | (unsigned)(uk1 >> 12)) & 0xfffu;
                    const unsigned longb = sideb | (ul0 < 15 ? 2u << ul0 : 0u) | (ul1 < 15 ? 2u << ul1 : 0u);
                    // ---- up  (levels in a gap between the two envs' ranges run idle: a range test here makes the compiler run the whole
                    // loop with per-lane exits and d in a VGPR)
                    for (int dctr = du0 > du1 ? du0 : du1; dctr > ulmin; --dctr) {
                        // (an opaque SCALAR copy of the level for everything the body compares with per-lane depths: where the body tests
                        // `turndep == d - 1`, value numbering rewrites the loop counter itself with the per-lane value it was found equal to, and
                        // the whole loop runs with its counter in a VGPR and per-lane exits - +20 VALU instructions per level)
                        int d = dctr;
                        asm volatile("" : "+s"(d));
                        V3 cn{0.f, 0.f, 0.f}, cf{0.f, 0.f, 0.f};
                        if (updep == d) {
                            const V3 na = aug * mul(Di, un_new);
                            const V3 fa = uf_new - V3{dot(col(E, 0), un_new), dot(col(E, 1), un_new), dot(col(E, 2), un_new)};
                            cn = na + cross(r, fa);
                            cf = fa;
                            un_new = V3{0.f, 0.f, 0.f};
                            uf_new = V3{0.f, 0.f, 0.f};
                            if (LIMITS) {  // the reaction of the joint's limit impulses: a pure torque on the parent
                                cn = cn - jt_new;
                                jt_new = V3{0.f, 0.f, 0.f};
                            }
                        }
                        if (!((longb >> d) & 1u)) {
                            // every link that hands over is the first child of its parent = the lane before it; the others hand over zeros:
                            // the shifted values are added as they are (DPP operand of the add)
                            un_new = un_new + from_next(cn);
                            uf_new = uf_new + from_next(cf);
                            un_tot = un_tot + from_next(cn);
                        } else {
                            // what arrives is kept apart: the link where a walk turns answers it with its Lambda
                            V3 rn = from_next(mask(firstchild, cn)), rf = from_next(mask(firstchild, cf));
                            if ((sideb >> d) & 1u) {
                                rn = rn + mask(has1, pull(cn, cl1)) + mask(has2, pull(cn, cl2));
                                rf = rf + mask(has1, pull(cf, cl1)) + mask(has2, pull(cf, cl2));
                            }
                            un_new = un_new + rn;
                            uf_new = uf_new + rf;
                            un_tot = un_tot + rn;
                            if (turndep == d - 1) {
                                Dw = Dw + mul(Lam.A, rn) + mul(Lam.B, rf);
                                Dv = Dv + V3{dot(col(Lam.B, 0), rn), dot(col(Lam.B, 1), rn), dot(col(Lam.B, 2), rn)} + mul(Lam.C, rf);
                            }
                        }
                    }
                    LLSUB(12);
                    // ---- down
                    const int dtop = dn0 > dn1 ? dn0 : dn1;
                    for (int d = dlmin + 1; d <= dtop; ++d) {
                        const bool nc = (nonchain >> d) & 1;
                        const V3 pdw = pp(Dw, nc), pdv = pp(Dv, nc);
                        if (dndep == d) {
                            const V3 tv = pdv + cross(pdw, r);
                            Dw = mul(Di, aug * pdw + un_tot) - mul(E, tv);
                            Dv = tv;
                        }
                    }
                    LLSUB(14);
                };
                // the walk returns to the root (its Lambda answers the total) and one root -> leaves pass moves the links down to depth dlast
                // (all of them, or those the sweep reads: the links down to the env's deepest stop)
                auto walk_close = [&](int dlast, bool all) {
                    const int i0 = ((int)__builtin_amdgcn_readlane(dep, cur0) << 4) | (M.side_depths[cur0] << 12);
                    const int i1 = ((int)__builtin_amdgcn_readlane(dep, 32 + cur1) << 4) | (M.side_depths[cur1] << 12);
                    const int slive = __builtin_amdgcn_readfirstlane((live0 ? 1 : 0) | (live1 ? 2 : 0));
                    walk_to(0, 0, i0, i1, (slive & 1) != 0, (slive & 2) != 0);
                    long long tsub = DIAG && a.prof ? clock64() : 0;
                    V3 ddw{0.f, 0.f, 0.f}, ddv{0.f, 0.f, 0.f};
                    if (lb == 0) {
                        ddw = Dw;
                        ddv = Dv;
                        w = w + ddw;
                        xd = xd + ddv;
                    }
                    for (int d = 1; d <= dlast; ++d) {
                        const bool nc = (nonchain >> d) & 1;
                        const V3 pdw = pp(ddw, nc), pdv = pp(ddv, nc);
                        if (dep == d && (all || insweep)) {
                            const V3 av = pdv + cross(pdw, r);
                            ddw = mul(Di, aug * pdw + un_tot) - mul(E, av);
                            ddv = av;
                            w = w + ddw;
                            xd = xd + ddv;
                        }
                    }
                    LLSUB(14);
                };
                // ---- ball x ground: a stop of its own after the last link - the last rows of a forward sweep, the first of a backward one (point
                // at -R z of the centre; rows n = z, t1 = x, t2 = y); returns whether any lane's rows changed something
                auto ball_ground_rows = [&](int done) -> int {
                    bool bmoved = false;
                    if (ballground && !((done >> half) & 1)) {
                        V3 bv{bl[BL_VEL], bl[BL_VEL + 1], bl[BL_VEL + 2]}, bw{bl[BL_ANG], bl[BL_ANG + 1], bl[BL_ANG + 2]};
                        const V3 rb{0.f, 0.f, -BP.radius};
                        float lamn = bl[BL_GLAM];
                        const float gbias = TGS ? fminf(rowbias(bl[BL_GGAP]), bl[BL_GREST]) : bl[BL_GBIAS];
#pragma unroll
                        for (int ax = 0; ax < 3; ++ax) {
                            const V3 dir = ax == 0 ? V3{0.f, 0.f, 1.f} : (ax == 1 ? V3{1.f, 0.f, 0.f} : V3{0.f, 1.f, 0.f});
                            const V3 jb = cross(rb, dir);
                            const float wii = BP.inv_mass + BP.inv_inertia * dot(jb, jb);
                            const float rel = dot(dir, bv) + dot(jb, bw) + (ax == 0 ? gbias : 0.f);
                            const float old = bl[BL_GLAM + ax];
                            float nl = old - rel * __builtin_amdgcn_rcpf(wii);
                            if (ax == 0) nl = fmaxf(nl, 0.f);
                            else { const float lim = BP.fric_ground * lamn; nl = fminf(fmaxf(nl, -lim), lim); }
                            const float dl = nl - old;
                            bl[BL_GLAM + ax] = nl;
                            if (ax == 0) lamn = nl;
                            bv = bv + (dl * BP.inv_mass) * dir;
                            bw = bw + (dl * BP.inv_inertia) * jb;
                            bmoved = bmoved || dl != 0.f;
                        }
                        bl[BL_VEL] = bv.x; bl[BL_VEL + 1] = bv.y; bl[BL_VEL + 2] = bv.z;
                        bl[BL_ANG] = bw.x; bl[BL_ANG + 1] = bw.y; bl[BL_ANG + 2] = bw.z;
                    }
                    const unsigned long long bm = __ballot(bmoved);
                    return ((unsigned)bm != 0u ? 1 : 0) | ((unsigned)(bm >> 32) != 0u ? 2 : 0);
                };
                // an env whose whole sweep changed nothing has reached the fixed point of its rows: it takes no part in the remaining sweeps,
                // whatever the env it shares the wave with still does.  (Its later sweeps change nothing: exactly so with ascending sweeps,
                // where the same moves re-derive the same velocities - the block updates they would take are saved; with the ALT experiment a
                // sweep in the other direction reaches the same links over other moves, i.e. with other rounding, and an env that kept
                // iterating for its wave partner's sake would depend on it.)
                int done = 0;
                for (int it = 0; it < P.n_iter; ++it) {
                    unsigned t0 = (done & 1) ? 0u : v0, t1 = (done & 2) ? 0u : v1;
                    int moved = 0;  // bit h: env h changed something in this sweep (wave-uniform)
                    const bool backward = ALT && (it & 1);  // (wave-uniform)
                    if (TGS && it > 0) {
                        // gaps advance with the normal velocity the points have after the previous sweep (touched links are current)
#pragma unroll
                        for (int c = 0; c < 4; ++c) { const V3 rc = CS.cr(c); CS.set_bias(c, CS.bias(c) + hs * (rc.y * w.x - rc.x * w.y + xd.z)); }
                        tgs_irem = 1.f / (h - (float)it * hs);
                        if (LIMITS && any64(limact)) {
                            // ... a limit row's with the joint rate of its DOF (the walk was closed down to the deepest stop: joint and parent are current)
                            const V3 pw = pp(w, true);
                            if (limact) {
                                const V3 om = mulT(q2mat(Q4{park[PARK_Q * 64], park[(PARK_Q + 1) * 64], park[(PARK_Q + 2) * 64], park[(PARK_Q + 3) * 64]}), w - pw);
                                lbias[0] += hs * lsgn[0] * om.x; lbias[1] += hs * lsgn[1] * om.y; lbias[2] += hs * lsgn[2] * om.z;
                            }
                        }
                        if (BALL) {
                            // ... the ball's rows with the relative normal velocity of the two contact points (the ball's own point is at -R n of its
                            // centre: its spin does not move it along n)
                            const V3 bv{bl[BL_VEL], bl[BL_VEL + 1], bl[BL_VEL + 2]};
                            if (ballground) bl[BL_GGAP] = bl[BL_GGAP] + hs * bv.z;
                            if (ballhit) {
#pragma unroll 1
                                for (int j = 0; j < NBREC; ++j) {
                                    lds_vfloat* rk = bl + BL_RK + 16 * j;
                                    if (rk[RK_A] == 0.f || !ball_rec_mine(j)) continue;
                                    const V3 n{rk[RK_N], rk[RK_N + 1], rk[RK_N + 2]}, rl{rk[RK_RL], rk[RK_RL + 1], rk[RK_RL + 2]};
                                    rk[RK_GAP] = rk[RK_GAP] + hs * dot(bv - xd - cross(w, rl), n);
                                }
                            }
                        }
                    }
                    if (BALL && backward) moved |= ball_ground_rows(done);  // (the last stop of a forward sweep is the first of a backward one)
                    while (t0 | t1) {
                        const int b0 = t0 ? (backward ? 31 - __clz(t0) : __ffs(t0) - 1) : -1, b1 = t1 ? (backward ? 31 - __clz(t1) : __ffs(t1) - 1) : -1;
                        t0 &= ~(b0 < 0 ? 0u : 1u << b0);
                        t1 &= ~(b1 < 0 ? 0u : 1u << b1);
                        if (DIAG && a.prof && ((a.prof_heavy ? blockIdx.x < 8u : (blockIdx.x & 63) == 0)) && lane == 0) atomicAdd((unsigned long long*)&a.prof[8], 1ull);
                        // (readfirstlane: wave-uniform by construction, and the compiler must know it - the level loops are scalar loops)
                        const int smv = __builtin_amdgcn_readfirstlane(((live0 && b0 >= 0 && b0 != cur0) ? 1 : 0) | ((live1 && b1 >= 0 && b1 != cur1) ? 2 : 0));
                        int came_down = 0;  // LIMITS: bit h = env h has just come DOWN to its link (the lowest common ancestor of the move lies above it)
                        if (smv) {
                            const int msel = backward ? minfo_rev : minfo;
                            const int i0 = __builtin_amdgcn_readlane(msel, b0 < 0 ? 0 : b0), i1 = __builtin_amdgcn_readlane(msel, 32 + (b1 < 0 ? 0 : b1));
                            walk_to(b0 < 0 ? 0 : b0, b1 < 0 ? 0 : b1, i0, i1, (smv & 1) != 0, (smv & 2) != 0);
                            if (LIMITS) came_down = smv & (((i0 & 15) < ((i0 >> 8) & 15) ? 1 : 0) | ((i1 & 15) < ((i1 >> 8) & 15) ? 2 : 0));
                        }
                        if (b0 >= 0) cur0 = b0;
                        if (b1 >= 0) cur1 = b1;
                        const int bsel = half ? b1 : b0;
                        if constexpr (LIMITS) {
                            // ---- the limit rows of the joint the walk stands on (before the contact rows of its link).  A limit impulse is a
                            // joint-space impulse: it joins the link's collected impulse, and its reaction, a pure torque, what the link hands up.
                            const bool hl0 = b0 >= 0 && ((lm0 >> b0) & 1u), hl1 = b1 >= 0 && ((lm1 >> b1) & 1u);
                            if (hl0 || hl1) {
                                // one joint up (the parent turns), one joint down again: the move that makes the joint and its parent current
                                const int m0i = __builtin_amdgcn_readlane(minfo, b0 < 0 ? 0 : b0), m1i = __builtin_amdgcn_readlane(minfo, 32 + (b1 < 0 ? 0 : b1));
                                const int d0 = (m0i >> 8) & 15, d1 = (m1i >> 8) & 15;
                                const int j0i = (d0 - 1) | (d0 << 4) | (d0 << 8) | ((((m0i >> 28) & 1) << d0) << 12);
                                const int j1i = (d1 - 1) | (d1 << 4) | (d1 << 8) | ((((m1i >> 28) & 1) << d1) << 12);
                                // the rows read the PARENT's velocity as well.  It is current when the walk came down through it; when the walk
                                // turned at this very link (or never left it) the parent still lacks what this link's subtree has collected since
                                // - this move hands exactly that up.
                                const int spre = __builtin_amdgcn_readfirstlane((((live0 && hl0) ? 1 : 0) | ((live1 && hl1) ? 2 : 0)) & ~came_down);
                                if (spre) walk_to(b0 < 0 ? 0 : b0, b1 < 0 ? 0 : b1, j0i, j1i, (spre & 1) != 0, (spre & 2) != 0);
                                const V3 wc = w + Dw;
                                const V3 pw = pp(wc, true);
                                bool lchg = false;
                                if (valid && lb == bsel && (half ? hl1 : hl0)) {
                                    const M3 R = q2mat(Q4{park[PARK_Q * 64], park[(PARK_Q + 1) * 64], park[(PARK_Q + 2) * 64], park[(PARK_Q + 3) * 64]});
                                    const V3 om0 = mulT(R, wc - pw);  // joint rate, body axes
                                    float om[3] = {om0.x, om0.y, om0.z}, tq[3] = {0.f, 0.f, 0.f};
#pragma unroll
                                    for (int i = 0; i < 3; ++i) {
                                        const V3 kc = i == 0 ? V3{Kd.xx, Kd.xy, Kd.xz} : (i == 1 ? V3{Kd.xy, Kd.yy, Kd.yz} : V3{Kd.xz, Kd.yz, Kd.zz});
                                        const float kii = i == 0 ? kc.x : (i == 1 ? kc.y : kc.z);
                                        const float rel = lsgn[i] * om[i] + rowbias(lbias[i]);
                                        const float nl = fmaxf(llam[i] - rel * __builtin_amdgcn_rcpf(kii), 0.f);
                                        const float dl = lsgn[i] != 0.f ? nl - llam[i] : 0.f;
                                        llam[i] += dl;
                                        const float sdl = lsgn[i] * dl;
                                        tq[i] = sdl;
                                        om[0] += kc.x * sdl; om[1] += kc.y * sdl; om[2] += kc.z * sdl;
                                    }
                                    const V3 jt = mul(R, V3{tq[0], tq[1], tq[2]});
                                    un_new = un_new + jt;
                                    un_tot = un_tot + jt;
                                    jt_new = jt_new + jt;
                                    lchg = tq[0] != 0.f || tq[1] != 0.f || tq[2] != 0.f;
                                }
                                const unsigned long long lc = __ballot((V2P_LL_EXP & 2) ? false : lchg);
                                const int sl = __builtin_amdgcn_readfirstlane(((unsigned)lc != 0u ? 1 : 0) | ((unsigned)(lc >> 32) != 0u ? 2 : 0));
                                if (sl) {
                                    // the joint answers, and so does everything above it: one joint up (the parent turns), one joint down again
                                    live0 = live0 || (sl & 1);
                                    live1 = live1 || (sl & 2);
                                    moved |= sl;
                                    // (deferring this move to the env's next one - it goes up through the parent anyway, or can be widened by a level when it
                                    // goes down into the joint's subtree - was built and measured in round 6: +0.4 %, inside the noise, because most limit
                                    // stops of the racket arm carry contact rows too and need the link current at once; not kept)
                                    if (!(V2P_LL_EXP & 1)) walk_to(b0 < 0 ? 0 : b0, b1 < 0 ? 0 : b1, j0i, j1i, (sl & 1) != 0, (sl & 2) != 0);
                                }
                            }
                        }
                        // ---- the rows of the link the walk stands on
                        long long tsub = DIAG && a.prof ? clock64() : 0;
                        const bool me = valid && lb == bsel && (!LIMITS || (((half ? m1 : m0) >> (bsel < 0 ? 0 : bsel)) & 1u));
                        V3 gn{0.f, 0.f, 0.f}, gf{0.f, 0.f, 0.f};  // what these rows add to the link's impulse
                        if (me) {
                            V3 wl = w + Dw, xl = xd + Dv;
                            // all four records of the link at once (one LDS round trip instead of one per point; unused slots hold zeros)
                            V3 rr4[4], lam4[4];
                            float bias4[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) { rr4[c] = CS.cr(c); lam4[c] = CS.lam(c); bias4[c] = CS.bias(c); }
                            V3 ws0{0.f, 0.f, 0.f}, xs0{0.f, 0.f, 0.f};  // VFRIC: v* of the link
                            if constexpr (VFRIC) { ws0 = park_get3(PARK_W0); xs0 = park_get3(PARK_XD0); }
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const bool active = c < cnt;
                                if (!any64(active)) break;  // uniform over the (at most two) touched links solved here
                                const V3 rr = rr4[c];
                                V3 t1{1.f, 0.f, 0.f}, t2{0.f, 1.f, 0.f};
                                if constexpr (VFRIC) vfric_frame(ws0, xs0, rr, t1, t2);
                                const V3 lam0 = lam4[c];
                                float ln = lam0.x, l1 = lam0.y, l2 = lam0.z;
                                // a point without normal impulse (hence without friction impulses: they are clamped to mu x normal)
                                // that is separating stays as it is: its three rows would change nothing
                                // (masked per lane as well, so that an env's numbers do not depend on what its wave partner does)
                                const float bias_c = rowbias(bias4[c]);
                                const bool act = active && !(ln == 0.f && rr.y * wl.x - rr.x * wl.y + xl.z + bias_c >= 0.f);
                                if (!any64(act)) continue;
#pragma unroll
                                for (int ax = 0; ax < 3; ++ax) {
                                    const V3 dir = ax == 0 ? V3{0.f, 0.f, 1.f} : (VFRIC ? (ax == 1 ? t1 : t2) : (ax == 1 ? V3{1.f, 0.f, 0.f} : V3{0.f, 1.f, 0.f}));
                                    const V3 jn = cross(rr, dir);
                                    const V3 yw = mul(Lam.A, jn) + mul(Lam.B, dir);
                                    const V3 yv = V3{dot(col(Lam.B, 0), jn), dot(col(Lam.B, 1), jn), dot(col(Lam.B, 2), jn)} + mul(Lam.C, dir);
                                    const float wii = dot(jn, yw) + dot(dir, yv);
                                    const float rel = dot(jn, wl) + dot(dir, xl) + (ax == 0 ? bias_c : 0.f);
                                    const float old = ax == 0 ? ln : (ax == 1 ? l1 : l2);
                                    float nl = old - rel * __builtin_amdgcn_rcpf(wii);
                                    if (ax == 0) nl = fmaxf(nl, 0.f);
                                    else { const float lim = P.mu * ln; nl = fminf(fmaxf(nl, -lim), lim); }
                                    const float dl = act ? nl - old : 0.f;
                                    if (ax == 0) ln += dl; else if (ax == 1) l1 += dl; else l2 += dl;
                                    wl = wl + dl * yw;
                                    xl = xl + dl * yv;
                                    gn = gn + dl * jn;
                                    gf = gf + dl * dir;
                                }
                                CS.set_lam(c, V3{ln, l1, l2});
                            }
                            if (BALL && ballhit) {
                                // ---- ball x racket points: two-body rows (ball point velocity minus racket point velocity); the ball side is
                                // a free sphere (1/m, 1/I), the link side goes through Lambda_b like every row of this block
                                V3 bv{bl[BL_VEL], bl[BL_VEL + 1], bl[BL_VEL + 2]}, bw{bl[BL_ANG], bl[BL_ANG + 1], bl[BL_ANG + 2]};
#pragma unroll 1
                                for (int j = 0; j < NBREC; ++j) {
                                    lds_vfloat* rk = bl + BL_RK + 16 * j;
                                    if (rk[RK_A] == 0.f || !ball_rec_mine(j)) continue;
                                    const V3 n{rk[RK_N], rk[RK_N + 1], rk[RK_N + 2]}, rl{rk[RK_RL], rk[RK_RL + 1], rk[RK_RL + 2]};
                                    V3 t1v, t2v;
                                    ball_dirs(n, t1v, t2v);
                                    const V3 rb = -BP.radius * n;
                                    float lamn = rk[RK_LAM];
                                    const float rkbias = TGS ? fminf(rowbias(rk[RK_GAP]), rk[RK_REST]) : rk[RK_BIAS];
#pragma unroll 1
                                    for (int ax = 0; ax < 3; ++ax) {
                                        const V3 dir = ax == 0 ? n : (ax == 1 ? t1v : t2v);
                                        const V3 jn = cross(rl, dir), jb = cross(rb, dir);
                                        const V3 yw = mul(Lam.A, jn) + mul(Lam.B, dir);
                                        const V3 yv = V3{dot(col(Lam.B, 0), jn), dot(col(Lam.B, 1), jn), dot(col(Lam.B, 2), jn)} + mul(Lam.C, dir);
                                        const float wii = dot(jn, yw) + dot(dir, yv) + BP.inv_mass + BP.inv_inertia * dot(jb, jb);
                                        const float rel = dot(dir, bv) + dot(jb, bw) - dot(jn, wl) - dot(dir, xl) + (ax == 0 ? rkbias : 0.f);
                                        const float old = rk[RK_LAM + ax];
                                        float nl = old - rel * __builtin_amdgcn_rcpf(wii);
                                        if (ax == 0) nl = fmaxf(nl, 0.f);
                                        else { const float lim = (j < 2 ? BP.fric_racket : BP.fric_body) * lamn; nl = fminf(fmaxf(nl, -lim), lim); }
                                        const float dl = nl - old;
                                        rk[RK_LAM + ax] = nl;
                                        if (ax == 0) lamn = nl;
                                        bv = bv + (dl * BP.inv_mass) * dir;      // +impulse on the ball
                                        bw = bw + (dl * BP.inv_inertia) * jb;
                                        wl = wl - dl * yw;                        // -impulse on the racket's link
                                        xl = xl - dl * yv;
                                        gn = gn - dl * jn;
                                        gf = gf - dl * dir;
                                    }
                                }
                                bl[BL_VEL] = bv.x; bl[BL_VEL + 1] = bv.y; bl[BL_VEL + 2] = bv.z;
                                bl[BL_ANG] = bw.x; bl[BL_ANG + 1] = bw.y; bl[BL_ANG + 2] = bw.z;
                            }
                            Dw = wl - w;
                            Dv = xl - xd;
                            un_tot = un_tot + gn;
                            un_new = un_new + gn;
                            uf_new = uf_new + gf;
                        }
                        const unsigned long long chg = __ballot(gn.x != 0.f || gn.y != 0.f || gn.z != 0.f || gf.x != 0.f || gf.y != 0.f || gf.z != 0.f);
                        live0 = live0 || (unsigned)chg != 0u;
                        live1 = live1 || (unsigned)(chg >> 32) != 0u;
                        moved |= ((unsigned)chg != 0u ? 1 : 0) | ((unsigned)(chg >> 32) != 0u ? 2 : 0);
                        if (DIAG && a.prof && !chg && ((a.prof_heavy ? blockIdx.x < 8u : (blockIdx.x & 63) == 0)) && lane == 0) atomicAdd((unsigned long long*)&a.prof[19], 1ull);
                        LLSUB(11);
                    }
                    if (BALL && !backward) moved |= ball_ground_rows(done);
                    if (TGS && (live0 || live1)) {
                        walk_close(dneed, false);  // (the links below catch up once, after the last iteration)
                        un_tot = un_new = uf_new = Dw = Dv = V3{0.f, 0.f, 0.f};
                        live0 = live1 = false;
                    }
                    if (!TGS) {  // (PGS: fixed biases - a sweep without any change would be repeated by the remaining ones)
                        done |= ~moved & 3;
                        if (done == 3) break;
                    }
                }
                if (!TGS && (live0 || live1)) walk_close(maxd, true);
                } else
                for (int it = 0; it < P.n_iter; ++it) {
                    static_assert(WALK || !(TGS && (BALL || LIMITS)), "the per-group propagation (V2P_LL_WALK=0) solves the ball and limit rows under PGS only");
                    unsigned t0 = m0, t1 = m1, l0 = lm0, l1 = lm1;
                    bool moved = false;
                    if (TGS && it > 0) {
                        // gaps advance with the normal velocity the points have after the previous sweep (touched links are current)
#pragma unroll
                        for (int c = 0; c < 4; ++c) { const V3 rc = CS.cr(c); CS.set_bias(c, CS.bias(c) + hs * (rc.y * w.x - rc.x * w.y + xd.z)); }
                        tgs_irem = 1.f / (h - (float)it * hs);
                    }
                    while (t0 | t1 | l0 | l1) {
                        // ---- one GROUP per env: a touched link and, while the next touched link (ascending order) is a child of the
                        // one just solved, that child too.  Inside a group a link sees its parent's impulses through the parent's own
                        // response (Lambda_parent x impulse, propagated over one joint), and the leaf->root->leaves propagation runs
                        // ONCE for the whole chain (a limb lying on the ground, ankle + toe of a standing foot) - linear, so the
                        // sequence of row updates is exactly the one of solving the links one by one.
                        int b0 = t0 ? __ffs(t0) - 1 : -1, b1 = t1 ? __ffs(t1) - 1 : -1;
                        // LIMITS: the next event of an env is the limit block of joint j when no touched link below j is left
                        const int j0 = l0 ? __ffs(l0) - 1 : 99, j1 = l1 ? __ffs(l1) - 1 : 99;
                        const bool lim0 = LIMITS && j0 != 99 && (b0 < 0 || j0 <= b0), lim1 = LIMITS && j1 != 99 && (b1 < 0 || j1 <= b1);
                        if (lim0) { b0 = j0; l0 &= l0 - 1; } else t0 &= t0 - 1;
                        if (lim1) { b1 = j1; l1 &= l1 - 1; } else t1 &= t1 - 1;
                        const bool mylim = half ? lim1 : lim0;
                        int last0 = b0, last1 = b1;
                        long long tsub = DIAG && a.prof ? clock64() : 0;
                        if (DIAG && a.prof && ((a.prof_heavy ? blockIdx.x < 8u : (blockIdx.x & 63) == 0)) && lane == 0) atomicAdd((unsigned long long*)&a.prof[8], 1ull);
                        V3 un{0.f, 0.f, 0.f}, uf{0.f, 0.f, 0.f};
                        V3 Dw{0.f, 0.f, 0.f}, Dv{0.f, 0.f, 0.f};  // velocity change of the link just solved due to the group's impulses so far
                        V3 jt{0.f, 0.f, 0.f};  // LIMITS: the joint impulse of this block, world axes (its reaction goes to the parent)
                        if (LIMITS && (lim0 || lim1)) {
                            const V3 pw = pp(w, true);  // all links are current between blocks
                            if (valid && mylim && lb == (half ? b1 : b0)) {
                                const M3 R = q2mat(Q4{park[PARK_Q * 64], park[(PARK_Q + 1) * 64], park[(PARK_Q + 2) * 64], park[(PARK_Q + 3) * 64]});
                                const V3 om0 = mulT(R, w - pw);  // joint rate, body axes
                                float om[3] = {om0.x, om0.y, om0.z}, tq[3] = {0.f, 0.f, 0.f};
#pragma unroll
                                for (int i = 0; i < 3; ++i) {
                                    const V3 kc = i == 0 ? V3{Kd.xx, Kd.xy, Kd.xz} : (i == 1 ? V3{Kd.xy, Kd.yy, Kd.yz} : V3{Kd.xz, Kd.yz, Kd.zz});
                                    const float kii = i == 0 ? kc.x : (i == 1 ? kc.y : kc.z);
                                    const float rel = lsgn[i] * om[i] + lbias[i];
                                    const float nl = fmaxf(llam[i] - rel * __builtin_amdgcn_rcpf(kii), 0.f);
                                    const float dl = lsgn[i] != 0.f ? nl - llam[i] : 0.f;
                                    llam[i] += dl;
                                    const float sdl = lsgn[i] * dl;
                                    tq[i] = sdl;
                                    om[0] += kc.x * sdl; om[1] += kc.y * sdl; om[2] += kc.z * sdl;
                                }
                                jt = mul(R, V3{tq[0], tq[1], tq[2]});
                                un = jt;
                            }
                        }
                        for (int step = 0;; ++step) {
                            const int bsel = half ? b1 : b0;
                            const bool me = valid && (lb == bsel) && !(LIMITS && mylim);
                            V3 tw{0.f, 0.f, 0.f}, tv{0.f, 0.f, 0.f};
                            if (step > 0) {
                                const V3 pdw = pp(Dw, true), pdv = pp(Dv, true);
                                if (me) {
                                    tv = pdv + cross(pdw, r);
                                    tw = mul(Di, aug * pdw) - mul(E, tv);
                                }
                            }
                            if (me) {
                                V3 wl = w + tw, xl = xd + tv;
#if V2P_LL_PREFETCH_ROWS
                                // all four records of the link at once (one LDS round trip instead of one per point; unused slots hold zeros)
                                V3 rr4[4], lam4[4];
                                float bias4[4];
#pragma unroll
                                for (int c = 0; c < 4; ++c) { rr4[c] = CS.cr(c); lam4[c] = CS.lam(c); bias4[c] = CS.bias(c); }
#endif
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    const bool active = c < cnt;
                                    if (!any64(active)) break;  // uniform over the (at most two) touched links solved here
#if V2P_LL_PREFETCH_ROWS
                                    V3 rr = rr4[c];
                                    const V3 lam0 = lam4[c];
#else
                                    V3 rr = CS.cr(c);
                                    const V3 lam0 = CS.lam(c);
#endif
                                    float ln = lam0.x, l1 = lam0.y, l2 = lam0.z;
                                    // a point without normal impulse (hence without friction impulses: they are clamped to mu x normal)
                                    // that is separating stays as it is: its three rows would change nothing
                                    // (masked per lane as well, so that an env's numbers do not depend on what its wave partner does)
#if V2P_LL_PREFETCH_ROWS
                                    const float bias_c = rowbias(bias4[c]);
#else
                                    const float bias_c = rowbias(CS.bias(c));
#endif
                                    const bool act = active && !(ln == 0.f && rr.y * wl.x - rr.x * wl.y + xl.z + bias_c >= 0.f);
                                    if (!any64(act)) continue;
#pragma unroll
                                    for (int ax = 0; ax < 3; ++ax) {
                                        V3 dir = ax == 0 ? V3{0.f, 0.f, 1.f} : (ax == 1 ? V3{1.f, 0.f, 0.f} : V3{0.f, 1.f, 0.f});
                                        V3 jn = cross(rr, dir);
                                        V3 yw = mul(Lam.A, jn) + mul(Lam.B, dir);
                                        V3 yv = V3{dot(col(Lam.B, 0), jn), dot(col(Lam.B, 1), jn), dot(col(Lam.B, 2), jn)} + mul(Lam.C, dir);
                                        float wii = dot(jn, yw) + dot(dir, yv);
                                        float rel = dot(jn, wl) + dot(dir, xl) + (ax == 0 ? bias_c : 0.f);
                                        float old = ax == 0 ? ln : (ax == 1 ? l1 : l2);
                                        float nl = old - rel * __builtin_amdgcn_rcpf(wii);
                                        if (ax == 0) nl = fmaxf(nl, 0.f);
                                        else { float lim = P.mu * ln; nl = fminf(fmaxf(nl, -lim), lim); }
                                        float dl = act ? nl - old : 0.f;
                                        if (ax == 0) ln += dl; else if (ax == 1) l1 += dl; else l2 += dl;
                                        wl = wl + dl * yw;
                                        xl = xl + dl * yv;
                                        un = un + dl * jn;
                                        uf = uf + dl * dir;
                                    }
                                    CS.set_lam(c, V3{ln, l1, l2});
                                }
                                if (BALL && ballhit) {
                                    // ---- ball x racket points: two-body rows (ball point velocity minus racket point velocity); the ball side is
                                    // a free sphere (1/m, 1/I), the link side goes through Lambda_b like every row of this block
                                    V3 bv{bl[BL_VEL], bl[BL_VEL + 1], bl[BL_VEL + 2]}, bw{bl[BL_ANG], bl[BL_ANG + 1], bl[BL_ANG + 2]};
#pragma unroll 1
                                    for (int j = 0; j < NBREC; ++j) {
                                        lds_vfloat* rk = bl + BL_RK + 16 * j;
                                        if (rk[RK_A] == 0.f || !ball_rec_mine(j)) continue;
                                        const V3 n{rk[RK_N], rk[RK_N + 1], rk[RK_N + 2]}, rl{rk[RK_RL], rk[RK_RL + 1], rk[RK_RL + 2]};
                                        V3 t1, t2;
                                        ball_dirs(n, t1, t2);
                                        const V3 rb = -BP.radius * n;
                                        float lamn = rk[RK_LAM];
#pragma unroll 1
                                        for (int ax = 0; ax < 3; ++ax) {
                                            const V3 dir = ax == 0 ? n : (ax == 1 ? t1 : t2);
                                            const V3 jn = cross(rl, dir), jb = cross(rb, dir);
                                            const V3 yw = mul(Lam.A, jn) + mul(Lam.B, dir);
                                            const V3 yv = V3{dot(col(Lam.B, 0), jn), dot(col(Lam.B, 1), jn), dot(col(Lam.B, 2), jn)} + mul(Lam.C, dir);
                                            const float wii = dot(jn, yw) + dot(dir, yv) + BP.inv_mass + BP.inv_inertia * dot(jb, jb);
                                            const float rel = dot(dir, bv) + dot(jb, bw) - dot(jn, wl) - dot(dir, xl) + (ax == 0 ? rk[RK_BIAS] : 0.f);
                                            const float old = rk[RK_LAM + ax];
                                            float nl = old - rel * __builtin_amdgcn_rcpf(wii);
                                            if (ax == 0) nl = fmaxf(nl, 0.f);
                                            else { const float lim = (j < 2 ? BP.fric_racket : BP.fric_body) * lamn; nl = fminf(fmaxf(nl, -lim), lim); }
                                            const float dl = nl - old;
                                            rk[RK_LAM + ax] = nl;
                                            if (ax == 0) lamn = nl;
                                            bv = bv + (dl * BP.inv_mass) * dir;      // +impulse on the ball
                                            bw = bw + (dl * BP.inv_inertia) * jb;
                                            wl = wl - dl * yw;                        // -impulse on the racket's link
                                            xl = xl - dl * yv;
                                            un = un - dl * jn;
                                            uf = uf - dl * dir;
                                        }
                                    }
                                    bl[BL_VEL] = bv.x; bl[BL_VEL + 1] = bv.y; bl[BL_VEL + 2] = bv.z;
                                    bl[BL_ANG] = bw.x; bl[BL_ANG + 1] = bw.y; bl[BL_ANG + 2] = bw.z;
                                }
                                Dw = wl - w;  // = tw + Lambda (un, uf): what this link's child (if it is next) starts from
                                Dv = xl - xd;
                            }
                            // does the chain go on?  (next touched link of the env, ascending, is a child of the one just solved)
                            const int n0 = t0 ? __ffs(t0) - 1 : -1, n1 = t1 ? __ffs(t1) - 1 : -1;
                            const bool c0 = !lim0 && b0 >= 0 && n0 >= 0 && ((chain0 >> n0) & 1u), c1 = !lim1 && b1 >= 0 && n1 >= 0 && ((chain1 >> n1) & 1u);
                            if (!(c0 || c1)) break;
                            b0 = c0 ? n0 : -1;
                            b1 = c1 ? n1 : -1;
                            if (c0) { t0 &= t0 - 1; last0 = n0; }
                            if (c1) { t1 &= t1 - 1; last1 = n1; }
                        }
                        const int blast = half ? last1 : last0;
                        const bool onpath = valid && blast >= 0 && ((desc >> blast) & 1);  // the group's deepest link or one of its ancestors
                        LLSUB(11);
                        // an update that changed no impulse (separated or saturated points) moves nothing: skip the propagation
                        if (!any64(un.x != 0.f || un.y != 0.f || un.z != 0.f || uf.x != 0.f || uf.y != 0.f || uf.z != 0.f)) {
                            if (DIAG && a.prof && ((a.prof_heavy ? blockIdx.x < 8u : (blockIdx.x & 63) == 0)) && lane == 0) atomicAdd((unsigned long long*)&a.prof[19], 1ull);
                            continue;
                        }
                        moved = true;
                        // ---- net impulse (un, uf) at the touched link: leaf -> root along the path, level by level
                        // (after its own level a path link's un is final: it is the joint-space impulse the way down needs)
                        for (int d = dneed; d >= 1; --d) {
                            if (!any64(onpath && dep == d)) continue;  // nothing to hand up from this level
                            V3 cn{0.f, 0.f, 0.f}, cf{0.f, 0.f, 0.f};
                            if (dep == d && onpath) {
                                V3 na = aug * mul(Di, un);
                                V3 fa = uf - V3{dot(col(E, 0), un), dot(col(E, 1), un), dot(col(E, 2), un)};
                                cn = na + cross(r, fa);
                                if (LIMITS) cn = cn - jt;
                                cf = fa;
                            }
                            if ((nonchain >> d) & 1) {
                                un = un + mask(has0, from_next(cn));
                                uf = uf + mask(has0, from_next(cf));
                            } else {
                                un = un + from_next(cn);
                                uf = uf + from_next(cf);
                            }
                            if (((multi >> d) & 1) && any64(dep == d && onpath && !firstchild)) {  // path enters its parent through child 1 or 2
                                un = un + mask(has1, pull(cn, cl1)) + mask(has2, pull(cn, cl2));
                                uf = uf + mask(has1, pull(cf, cl1)) + mask(has2, pull(cf, cl2));
                            }
                        }
                        LLSUB(12);
                        // root response, then root -> leaves: every link moves
                        V3 ddw{0.f, 0.f, 0.f}, ddv{0.f, 0.f, 0.f};
                        if (lb == 0) {
                            ddw = mul(Lam.A, un) + mul(Lam.B, uf);
                            ddv = V3{dot(col(Lam.B, 0), un), dot(col(Lam.B, 1), un), dot(col(Lam.B, 2), un)} + mul(Lam.C, uf);
                            w = w + ddw;
                            xd = xd + ddv;
                        }
                        for (int d = 1; d <= dneed; ++d) {
                            const bool nc = (nonchain >> d) & 1;
#if V2P_LL_DPP_DOWN
                            V3 pdw = pp.fast(ddw, nc), pdv = pp.fast(ddv, nc);
#else
                            V3 pdw = pp(ddw, nc), pdv = pp(ddv, nc);
#endif
                            if (dep == d && insweep) {
                                V3 av = pdv + cross(pdw, r);
                                ddw = mul(Di, aug * pdw + un) - mul(E, av);
                                ddv = av;
                                w = w + ddw;
                                xd = xd + ddv;
                            }
                        }
                        LLSUB(14);
                    }
                    if (BALL) {
                        // ---- ball x ground: the last rows of the iteration (point at -R z of the centre; rows n = z, t1 = x, t2 = y)
                        bool bmoved = false;
                        if (ballground) {
                            V3 bv{bl[BL_VEL], bl[BL_VEL + 1], bl[BL_VEL + 2]}, bw{bl[BL_ANG], bl[BL_ANG + 1], bl[BL_ANG + 2]};
                            const V3 rb{0.f, 0.f, -BP.radius};
                            float lamn = bl[BL_GLAM];
#pragma unroll
                            for (int ax = 0; ax < 3; ++ax) {
                                const V3 dir = ax == 0 ? V3{0.f, 0.f, 1.f} : (ax == 1 ? V3{1.f, 0.f, 0.f} : V3{0.f, 1.f, 0.f});
                                const V3 jb = cross(rb, dir);
                                const float wii = BP.inv_mass + BP.inv_inertia * dot(jb, jb);
                                const float rel = dot(dir, bv) + dot(jb, bw) + (ax == 0 ? bl[BL_GBIAS] : 0.f);
                                const float old = bl[BL_GLAM + ax];
                                float nl = old - rel * __builtin_amdgcn_rcpf(wii);
                                if (ax == 0) nl = fmaxf(nl, 0.f);
                                else { const float lim = BP.fric_ground * lamn; nl = fminf(fmaxf(nl, -lim), lim); }
                                const float dl = nl - old;
                                bl[BL_GLAM + ax] = nl;
                                if (ax == 0) lamn = nl;
                                bv = bv + (dl * BP.inv_mass) * dir;
                                bw = bw + (dl * BP.inv_inertia) * jb;
                                bmoved = bmoved || dl != 0.f;
                            }
                            bl[BL_VEL] = bv.x; bl[BL_VEL + 1] = bv.y; bl[BL_VEL + 2] = bv.z;
                            bl[BL_ANG] = bw.x; bl[BL_ANG + 1] = bw.y; bl[BL_ANG + 2] = bw.z;
                        }
                        if (any64(bmoved)) moved = true;
                    }
                    if (!TGS && !moved) break;  // a whole iteration without any change: the remaining ones would repeat it (PGS: fixed biases)
                }
                // links below the deepest touched one: one catch-up pass with the accumulated motion of their parents
                // (PGS: the walk moves every link when it closes)
                V3 accw = w - park_get3(PARK_W0), accv = xd - park_get3(PARK_XD0);
                for (int d = dmin + 1; d <= ((WALK && !TGS) ? 0 : maxd); ++d) {
                    const bool nc = (nonchain >> d) & 1;
                    V3 pdw = pp(accw, nc), pdv = pp(accv, nc);
                    if (dep == d && !insweep) {
                        V3 av = pdv + cross(pdw, r);
                        accw = mul(Di, aug * pdw) - mul(E, av);
                        accv = av;
                        w = w + accw;
                        xd = xd + accv;
                    }
                }
            }
        }

        LLPH(6);
        V3 cimp_v{0.f, 0.f, 0.f};  // VFRIC: net contact impulse of this link's ground points, world axes (v* is about to leave its slots)
        if constexpr (VFRIC && CONTACT) {
            const V3 ws0 = park_get3(PARK_W0), xs0 = park_get3(PARK_XD0);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < cnt) {
                    V3 t1, t2;
                    vfric_frame(ws0, xs0, CS.cr(c), t1, t2);
                    const V3 lc = CS.lam(c);
                    cimp_v = cimp_v + V3{lc.y * t1.x + lc.z * t2.x, lc.y * t1.y + lc.z * t2.y, lc.x};
                }
        }
        q = Q4{park[PARK_Q * 64], park[(PARK_Q + 1) * 64], park[(PARK_Q + 2) * 64], park[(PARK_Q + 3) * 64]};
        x = park_get3(PARK_X);
        // ================================================================ velocities -> generalized, damping, clamp, integrate
        const float sc = 1.f / (1.f + h * P.ang_damp);
        const float wmax = P.max_ang_vel;
        {
            Q4 pq = pp(q, true);
            V3 pw = pp(w, true);
            if (b != 0) {
                V3 wn = mulT(q2mat(q), w - pw);               // joint rate, body axes (undamped)
                Q4 jold = qnormalize(qmul(qconj(pq), q));  // joint quaternion of the old configuration
                if (last && valid && live_env && !frozen) {  // joint drive torque actually applied over the substep (implicit form)
                    const float kp = S->kp[b], kd = S->kd[b];
                    V3 tf = kp * (park_get3(PARK_TAR) - quat_to_expmap_stable(jold) - h * wn) - kd * wn;
                    int bh = b;
                    asm volatile("" : "+v"(bh));
                    float* of = a.x_dof_force + env_here() * NDOF + 3 * (bh - 1);
                    of[0] = tf.x; of[1] = tf.y; of[2] = tf.z;
                }
                wn = sc * wn;
                float n2 = dot(wn, wn);
                if (n2 > wmax * wmax) wn = (wmax * rsqrtf(n2)) * wn;
                wt = wn;
                jq = qnormalize(qmul(jold, rotvec_to_quat(h * wn)));  // body-frame rate: right multiply
            } else {
                V3 w0 = sc * w;
                float n2 = dot(w0, w0);
                if (n2 > wmax * wmax) w0 = (wmax * rsqrtf(n2)) * w0;
                w = w0;
                x = x + h * xd;
                q = qnormalize(qmul(rotvec_to_quat(h * w0), q));  // world-frame rate: left multiply
            }
        }
        if (BALL) {
            auto ball_rec_mine_out = [&](int j) -> bool { return j < 2 ? lb == BP.racket_link : lb == (int)bl[BL_RK + 16 * j + RK_LINK]; };
            // force on the ball in this substep from the racket (sum over the two cylinders) and from the hull points, world axes
            V3 frk{0.f, 0.f, 0.f}, fbd{0.f, 0.f, 0.f}, fmine{0.f, 0.f, 0.f};  // from the racket, from all hulls, from this lane's hull
            {
                const float ih = PHYS_RCP(h);
#pragma unroll
                for (int j = 0; j < NBREC; ++j) {
                    lds_vfloat* rk = bl + BL_RK + 16 * j;
                    if (rk[RK_A] != 0.f && (ball_lane || (valid && ball_rec_mine_out(j)))) {
                        const V3 n{rk[RK_N], rk[RK_N + 1], rk[RK_N + 2]};
                        V3 t1, t2;
                        t1 = cross(n, V3{0.f, 0.f, 1.f});
                        if (dot(t1, t1) < 1e-6f) t1 = cross(n, V3{1.f, 0.f, 0.f});
                        t1 = rsqrtf(dot(t1, t1)) * t1;
                        t2 = cross(n, t1);
                        const V3 f = ih * (rk[RK_LAM] * n + rk[RK_LAM + 1] * t1 + rk[RK_LAM + 2] * t2);
                        if (j < 2) frk = frk + f;
                        else { fbd = fbd + f; if (!ball_lane) fmine = fmine + f; }
                    }
                }
            }
            if (ball_lane) {
                // ---- the ball: angular damping, clamp, integrate; outputs after the last substep of every simulate() call
                V3 bp{bl[BL_POS], bl[BL_POS + 1], bl[BL_POS + 2]}, bv{bl[BL_VEL], bl[BL_VEL + 1], bl[BL_VEL + 2]}, bw{bl[BL_ANG], bl[BL_ANG + 1], bl[BL_ANG + 2]};
                Q4 bq{bl[BL_QUAT], bl[BL_QUAT + 1], bl[BL_QUAT + 2], bl[BL_QUAT + 3]};
                bw = PHYS_RCP(1.f + h * BP.ang_damp) * bw;
                const float n2 = dot(bw, bw);
                if (n2 > BP.max_ang_vel * BP.max_ang_vel) bw = (BP.max_ang_vel * rsqrtf(n2)) * bw;
                bp = bp + h * bv;
                bq = qnormalize(qmul(rotvec_to_quat(h * bw), bq));
                bl[BL_POS] = bp.x; bl[BL_POS + 1] = bp.y; bl[BL_POS + 2] = bp.z;
                bl[BL_QUAT] = bq.x; bl[BL_QUAT + 1] = bq.y; bl[BL_QUAT + 2] = bq.z; bl[BL_QUAT + 3] = bq.w;
                bl[BL_ANG] = bw.x; bl[BL_ANG + 1] = bw.y; bl[BL_ANG + 2] = bw.z;
                // (output addresses from the opaque env index: formed from `e` they are loop invariants, hoisted in front of the substep
                // loop and kept - spilled - across all of it: 70 dwords of scratch per lane in the racket + ball kernels)
                const int64_t e = env_here();
                if (live_env && !replay && sub % BP.sub_per_sim == BP.sub_per_sim - 1) {
                    const int ks = sub / BP.sub_per_sim, nsim = nsub / BP.sub_per_sim;
                    float* o = BP.per_sim + (e * nsim + ks) * 13;
                    o[0] = bp.x; o[1] = bp.y; o[2] = bp.z; o[3] = bq.x; o[4] = bq.y; o[5] = bq.z; o[6] = bq.w;
                    o[7] = bv.x; o[8] = bv.y; o[9] = bv.z; o[10] = bw.x; o[11] = bw.y; o[12] = bw.z;
                    const bool hit = frk.x != 0.f || frk.y != 0.f || frk.z != 0.f;
                    BP.hit_per_sim[e * nsim + ks] = hit ? 1 : 0;
                    if (BP.has_hit) {  // the reference's poll of the net contact forces after every simulate() call (:773-779)
                        if (ks == 0) flag_st(&BP.has_hit_now[e], 0);
                        if (BP.poll_hits && hit && !flag_ld(&BP.has_hit[e])) { flag_st(&BP.has_hit[e], 1); flag_st(&BP.has_hit_now[e], 1); }
                    }
                }
                if (last && live_env) {
                    const float ih = PHYS_RCP(h);
                    float* oc = BP.contact + e * 6;
                    oc[0] = frk.x; oc[1] = frk.y; oc[2] = frk.z;
                    oc[3] = bl[BL_GA] != 0.f ? bl[BL_GLAM + 1] * ih : 0.f; oc[4] = bl[BL_GA] != 0.f ? bl[BL_GLAM + 2] * ih : 0.f;
                    oc[5] = bl[BL_GA] != 0.f ? bl[BL_GLAM] * ih : 0.f;
                    if (BP.body_contact) { float* ob = BP.body_contact + e * 3; ob[0] = fbd.x; ob[1] = fbd.y; ob[2] = fbd.z; }
                }
            }
            if ((last || (BP.contact_sum && sub % BP.sub_per_sim == BP.sub_per_sim - 1)) && valid && live_env && !frozen) {
                // the reaction on the touched link enters its net contact force below
                const V3 fr = mask(lb == BP.racket_link, frk) + fmine;
                park[PARK_W0 * 64] = fr.x; park[(PARK_W0 + 1) * 64] = fr.y; park[(PARK_W0 + 2) * 64] = fr.z;
            }
        }
        // (racket + ball, opt-in: `_contact_forces_sum`, the net contact forces summed over the simulate() calls of a control step,
        // humanoid_smpl_im_mvae.py:781 - what refresh_net_contact_force_tensor shows after EVERY call is needed then, not only after the last)
        // One slot per simulate() call (`contact_part`, engine-owned), written by whoever runs the call's last substep - a replaying job
        // included: the same bits whoever writes them, so the order in which a pair's jobs run cannot matter - and added up by the job of the
        // env's LAST substep, which either waited for the jobs before it (their stores were drained in front of the progress word) or has
        // just replayed them itself.  (Round 3 accumulated in place, load + add + store per call: a job that ran ahead of a late predecessor
        // added onto a stale value, and the predecessor's own store then dropped what had been added.)
        const bool sum_now = BALL && BP.contact_sum && sub % BP.sub_per_sim == BP.sub_per_sim - 1;
        if ((last || sum_now) && valid && live_env && !frozen) {
            // net contact force per body = sum of impulses / h  (refresh_net_contact_force_tensor)
            V3 cforce{0.f, 0.f, 0.f};
            if (CONTACT) {
                const float ih = 1.f / h;
                if constexpr (VFRIC) cforce = ih * cimp_v;
                else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < cnt) { const V3 lc = CS.lam(c); cforce.z += lc.x * ih; cforce.x += lc.y * ih; cforce.y += lc.z * ih; }
                }
                if (BALL) cforce = cforce - park_get3(PARK_W0);
            }
            if (last) {
                float* oc = a.x_contact + (env_here() * NB + b) * 3;
                oc[0] = cforce.x; oc[1] = cforce.y; oc[2] = cforce.z;
            }
            if (sum_now) {  // (system-scope accesses: the calls of one step may run in different workgroups)
                const int ks = sub / BP.sub_per_sim, nsim = nsub / BP.sub_per_sim;
                float* const part = BP.contact_part + ((env_here() * nsim) * NB + b) * 3;
                if (!last) {
                    float* op = part + (int64_t)ks * NB * 3;
                    cstore(op, cforce.x); cstore(op + 1, cforce.y); cstore(op + 2, cforce.z);
                } else {
                    V3 tot{0.f, 0.f, 0.f};
                    for (int k = 0; k < ks; ++k) {
                        const float* ip = part + (int64_t)k * NB * 3;
                        tot = tot + V3{cload(ip), cload(ip + 1), cload(ip + 2)};
                    }
                    tot = tot + cforce;
                    float* os = BP.contact_sum + (env_here() * NB + b) * 3;
                    os[0] = tot.x; os[1] = tot.y; os[2] = tot.z;
                }
            }
        }
    }

    LLPH(7);
    // (opaque copy of the link index for everything after the substeps: indices and addresses formed from it are computed here instead of
    // being kept - spilled to scratch, which is HBM write traffic for every wave - since the prologue)
    int bo2 = b;
    asm volatile("" : "+v"(bo2));
    if (frozen && nsub > 0) {  // frozen env sharing a wave with a live one: it was carried along, its result is dropped
        if (b == 0) {
            q = Q4{st[SIDX(ST_ROOT_QUAT + 0)], st[SIDX(ST_ROOT_QUAT + 1)], st[SIDX(ST_ROOT_QUAT + 2)], st[SIDX(ST_ROOT_QUAT + 3)]};
            x = V3{st[SIDX(ST_ROOT_POS + 0)], st[SIDX(ST_ROOT_POS + 1)], st[SIDX(ST_ROOT_POS + 2)]};
            xd = V3{st[SIDX(ST_VEL + 0)], st[SIDX(ST_VEL + 1)], st[SIDX(ST_VEL + 2)]};
            w = V3{st[SIDX(ST_VEL + 3)], st[SIDX(ST_VEL + 4)], st[SIDX(ST_VEL + 5)]};
        } else {
            const int jb = ST_JQUAT + 4 * (bo2 - 1), vb = ST_VEL + 6 + 3 * (bo2 - 1);
            jq = Q4{st[SIDX(jb + 0)], st[SIDX(jb + 1)], st[SIDX(jb + 2)], st[SIDX(jb + 3)]};
            wt = V3{st[SIDX(vb + 0)], st[SIDX(vb + 1)], st[SIDX(vb + 2)]};
        }
    }
    // ==================================================================== final kinematics -> state, rigid-body state, dof_pos
    // (JOBS: only the job of the last substep produces the exposed tensors and the pairing keys; the others hand the state over)
    const V3 lpos{S->local_pos[bo2][0], S->local_pos[bo2][1], S->local_pos[bo2][2]};
#if V2P_LL_KIN_JUMP
    if (last_job) {
        V3 rr{0.f, 0.f, 0.f}, wrel{0.f, 0.f, 0.f};
        kin_jump(q, x, w, xd, jq, wt, lpos, rr, wrel);
    }
#else
    for (int d = 1; d <= (last_job ? maxd : 0); ++d) {
        const bool nc = (nonchain >> d) & 1;
        Q4 pq = pp(q, nc);
        V3 px = pp(x, nc), pw = pp(w, nc), pxd = pp(xd, nc);
        if (dep == d) {
            q = qnormalize(qmul(pq, jq));
            V3 rr = mul(q2mat(pq), lpos);
            x = px + rr;
            w = pw + mul(q2mat(q), wt);
            xd = pxd + cross(pw, rr);
        }
    }
#endif
    if (DIAG && a.wave_times && lane == 0) {
        long long* wt = a.wave_times + ((int64_t)blockIdx.x * LL_WPB + (threadIdx.x >> 6)) * 4;
        wt[0] = wt0; wt[1] = wall_clock64(); wt[2] = ksum * 8 + kdep + 1024 * (long long)tsum + 1048576ll * tmaxs + 1073741824ll * key_pred;
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); wt[3] = hw;
    }
    if (a.pair_hist && last_job) {
        // ---- pairing key of this env (touched links, then the deepest of them: what the sweep's cost follows) into its load bin;
        // the workgroup that finishes last turns the histogram into the first rank of every bin, what the next launch's lookup starts from
        // links about to touch (bounding box within the contact offset of the ground after one more control step at the current
        // vertical speed) count like touched ones: a humanoid that falls goes from 2-4 to 12+ touched links within one control
        // step, and a heavy pair that is dispatched late because it was predicted light is the tail of the whole launch (+17 %)
        int ksoon = 0;
        {
            const float rz0 = 2.f * (q.x * q.z - q.w * q.y), rz1 = 2.f * (q.y * q.z + q.w * q.x), rz2 = 1.f - 2.f * (q.x * q.x + q.y * q.y);
            const float zl = x.z + rz0 * S->aabb_c[bo2][0] + rz1 * S->aabb_c[bo2][1] + rz2 * S->aabb_c[bo2][2] -
                             (fabsf(rz0) * S->aabb_e[bo2][0] + fabsf(rz1) * S->aabb_e[bo2][1] + fabsf(rz2) * S->aabb_e[bo2][2]);
            const unsigned long long nb2 = __ballot(valid && zl + fminf(xd.z, 0.f) * P.dt < P.contact_offset);
            ksoon = __popc(half ? (unsigned)(nb2 >> 32) : (unsigned)nb2);
        }
        if (lb == 0 && live_env) {
            int key = frozen ? 0 : (ksum > ksoon ? ksum : ksoon) * 8 + (kdep > 7 ? 7 : kdep);
            key = key > PAIR_BINS - 1 ? PAIR_BINS - 1 : key;
            const int pos = atomicAdd(&a.pair_hist[PAIR_BINS - 1 - key], 1);
            a.pair_key[e] = key;
            a.pair_pos[e] = pos;
            a.pl_list_next[(int64_t)(PAIR_BINS - 1 - key) * N + pos] = (int32_t)e;  // what the next launch looks its envs up in
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's bin counts are in before its ticket is drawn
        int ticket = -1;
        if (lane == 0) ticket = atomicAdd(a.pair_done, 1);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket == nblk * LL_WPB - 1) {
            int c[4], mine = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { c[k] = __hip_atomic_load(&a.pair_hist[4 * lane + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); mine += c[k]; }
            int inc = mine;
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
            int ex = inc - mine;
#pragma unroll
            for (int k = 0; k < 4; ++k) { a.pair_start[4 * lane + k] = ex; ex += c[k]; a.pair_hist[4 * lane + k] = 0; }
            if (lane == 0) *a.pair_done = 0;
        }
    }
    // the output addresses are formed here, from an index the compiler cannot trace back to the prologue: computed up front they were
    // kept across the whole kernel as spilled 64-bit pointers (36 B of scratch per lane = 9 MB of HBM traffic per launch)
    int64_t e_out = e;
    asm volatile("" : "+v"(e_out));
    {
    const int64_t e = e_out;
    if constexpr (JOBS) if (!last_job) {
        // ---- hand the state over to the job of the next substep: system-scope stores, drained, then the progress word of the pair
        {
            // Link b owns the adjacent chunks 2b, 2b + 1 (one 32-byte read per lane on the other side).  Written by their owner, every
            // store instruction would touch half of each 64-byte line (PMC WRITE_SIZE: +14 MB per launch of partial-line write-throughs);
            // instead lane l of an env writes chunk 24 k + l in instruction k = 0, 1 - 24 adjacent chunks = six full lines -, fetching
            // the values from the lane that owns them (link 12 k + l / 2, its first or second chunk).
            const bool root = b == 0;
            const float A0 = root ? q.x : jq.x, A1 = root ? q.y : jq.y, A2 = root ? q.z : jq.z, A3 = root ? q.w : jq.w;
            const float B0 = root ? x.x : wt.x, B1 = root ? x.y : wt.y, B2 = root ? x.z : wt.z, B3 = root ? xd.x : 0.f;
            float* const ho = a.job_hand + ((int64_t)sjob * N + (V2P_LL_FAST_HANDOVER != 0 ? slot : e)) * HAND_FLOATS;
            const bool second = lb & 1;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int src = base + 12 * k + ((lb >> 1) < 12 ? (lb >> 1) : 0);
                const float a0 = pull(A0, src), a1 = pull(A1, src), a2 = pull(A2, src), a3 = pull(A3, src);
                const float b0 = pull(B0, src), b1 = pull(B1, src), b2 = pull(B2, src), b3 = pull(B3, src);
                if (valid && live_env) cstore4(ho + 4 * (24 * k + lb), second ? b0 : a0, second ? b1 : a1, second ? b2 : a2, second ? b3 : a3);
            }
            if (root && valid && live_env) {
                cstore4(ho + 4 * 48, xd.y, xd.z, w.x, w.y);
                if (a.actions && jstart(sjob + 1) < a.p.hold_sub) {  // the next job still applies the residual wrench: it travels along
                    const float* wp = park_all + (threadIdx.x >> 6) * LDS_FLOATS_PER_WAVE + base;
                    cstore4(ho + 4 * 49, w.z, wp[PARK_TAR * 64], wp[(PARK_TAR + 1) * 64], wp[(PARK_TAR + 2) * 64]);
                    cstore4(ho + 4 * 54, wp[25 + PARK_TAR * 64], wp[25 + (PARK_TAR + 1) * 64], wp[25 + (PARK_TAR + 2) * 64], 0.f);
                } else {
                    cstore4(ho + 4 * 49, w.z, 0.f, 0.f, 0.f);
                }
            }
        }
        if (BALL && ball_lane && live_env) {
            float* const ho = a.job_hand + ((int64_t)sjob * N + (V2P_LL_FAST_HANDOVER != 0 ? slot : e)) * HAND_FLOATS;
            cstore4(ho + 4 * 50, bl[0], bl[1], bl[2], bl[3]);
            cstore4(ho + 4 * 51, bl[4], bl[5], bl[6], bl[7]);
            cstore4(ho + 4 * 52, bl[8], bl[9], bl[10], bl[11]);
            cstore4(ho + 4 * 53, bl[12], bl[13], bl[14], bl[15]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (inline asm: the compiler must not drop or move this drain)
        if (lane == 0) {
            int bid_here = bid;  // (the address of the progress word is formed here: kept from the prologue it is a spilled 64-bit pointer)
            asm volatile("" : "+s"(bid_here));
            // (max, not store: a job that is overtaken - its successor gave up waiting and has already handed over further - must not turn
            // the word back)
            __hip_atomic_fetch_max(a.job_progress + (bid_here * LL_WPB + (threadIdx.x >> 6)), a.job_epoch * (a.p.nsub + 1) + sjob + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        timeline();
        return;
    }
    V3 qe{0.f, 0.f, 0.f};  // exposed dof position of this lane's joint
    if (valid && live_env) {
        if (b == 0) {
            st[SIDX(ST_ROOT_QUAT + 0)] = q.x; st[SIDX(ST_ROOT_QUAT + 1)] = q.y; st[SIDX(ST_ROOT_QUAT + 2)] = q.z; st[SIDX(ST_ROOT_QUAT + 3)] = q.w;
            st[SIDX(ST_ROOT_POS + 0)] = x.x; st[SIDX(ST_ROOT_POS + 1)] = x.y; st[SIDX(ST_ROOT_POS + 2)] = x.z;
            st[SIDX(ST_VEL + 0)] = xd.x; st[SIDX(ST_VEL + 1)] = xd.y; st[SIDX(ST_VEL + 2)] = xd.z;
            st[SIDX(ST_VEL + 3)] = w.x; st[SIDX(ST_VEL + 4)] = w.y; st[SIDX(ST_VEL + 5)] = w.z;
        } else {
            const int jb = ST_JQUAT + 4 * (bo2 - 1), vb = ST_VEL + 6 + 3 * (bo2 - 1);
            st[SIDX(jb + 0)] = jq.x; st[SIDX(jb + 1)] = jq.y; st[SIDX(jb + 2)] = jq.z; st[SIDX(jb + 3)] = jq.w;
            st[SIDX(vb + 0)] = wt.x; st[SIDX(vb + 1)] = wt.y; st[SIDX(vb + 2)] = wt.z;
            // exposed dof state: (exp-map position, joint rate) interleaved like gym's dof state tensor
            qe = quat_to_expmap_stable(jq);
            float* od = a.x_dof + (e * NDOF + 3 * (bo2 - 1)) * 2;
            od[0] = qe.x; od[1] = wt.x; od[2] = qe.y; od[3] = wt.y; od[4] = qe.z; od[5] = wt.z;
        }
        if (b == 0) {
            float* orr = a.x_root + e * 13;
            orr[0] = x.x; orr[1] = x.y; orr[2] = x.z;
            orr[3] = q.x; orr[4] = q.y; orr[5] = q.z; orr[6] = q.w;
            orr[7] = xd.x; orr[8] = xd.y; orr[9] = xd.z;
            orr[10] = w.x; orr[11] = w.y; orr[12] = w.z;
        }
    }
    if (BALL && live_env) {
        if (ball_lane) {
#pragma unroll
            for (int k = 0; k < 13; ++k) BP.state[e * 13 + k] = bl[k];
        }
        if (valid && lb == BP.racket_link) {  // rigid body 24 of the reference's tensor: the racket frame, welded to this link
            const V3 off = mul(q2mat(q), V3{BP.racket_off[0], BP.racket_off[1], BP.racket_off[2]});
            const V3 rx = x + off, rv = xd + cross(w, off);
            float* o = BP.racket_state + e * 13;
            o[0] = rx.x; o[1] = rx.y; o[2] = rx.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
            o[7] = rv.x; o[8] = rv.y; o[9] = rv.z; o[10] = w.x; o[11] = w.y; o[12] = w.z;
        }
    }
    // rigid-body state [24][13] of the env: a lane's row is 13 dwords at a 52-byte stride, which as 13 scalar stores per lane left the
    // L2 with partial lines from two XCDs' worth of neighbours (PMC WRITE_SIZE 1.7x the bytes).  The rows are staged through the
    // (now idle) LDS parking area and leave as 78 contiguous 16-byte stores per env: full lines.
    {
        float* const stage = park_all + (threadIdx.x >> 6) * LDS_FLOATS_PER_WAVE + half * (NB * 13);
        if (valid) {
            float* o = stage + b * 13;
            o[0] = x.x; o[1] = x.y; o[2] = x.z;
            o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
            o[7] = xd.x; o[8] = xd.y; o[9] = xd.z;
            o[10] = w.x; o[11] = w.y; o[12] = w.z;
        }
        if (live_env) {
            const float4* src = (const float4*)stage;
            float4* dst = (float4*)(a.x_rb + e * NB * 13);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int j = lb + 32 * r;
                if (j < NB * 13 / 4) dst[j] = src[j];
            }
        }
    }
    if constexpr (JOBS && !DIAG) {
        // ---- post-physics fused in (v2p_env_step): the same functions, compiled with the same semantics, as env_post_kernel; lane = body
        // holds exactly the values the exposed tensors were just given.  The post-physics kernel cost a launch of its own right behind
        // the tail of this one (the heaviest env pairs finish last, on an almost empty GPU); here every pair does it as it finishes.
        if (a.post.on) {
            const PostArgs& Z = a.post;
            bool fl = false;
            int64_t mid = 0;
            float t_new = 0.f;
            strict::RewardPartial rp{0.f, 0.f, 0.f, 0.f};
            if (valid && live_env) {
                mid = Z.motion_id[e];
                t_new = Z.b.cur_time[e] + P.dt;  // _cur_ref_motion_times += dt
                rp = strict::post_body(Z.b, Z.t, P, mid, t_new, Z.cur, e, b, strict::V3{x.x, x.y, x.z}, strict::Q4{q.x, q.y, q.z, q.w}, strict::V3{xd.x, xd.y, xd.z},
                                       strict::V3{w.x, w.y, w.z}, strict::V3{qe.x, qe.y, qe.z}, strict::V3{wt.x, wt.y, wt.z}, fl, V2P_LL_TARGET_HEAD == 0);
            }
            // reward sums over the bodies, bodies ascending like env_post_kernel: the terms go through this lane's LDS column (the parking
            // area is idle by now; a wave's LDS accesses execute in order), lanes 0..3 of each env add one term each
            park[0] = rp.dof; park[64] = rp.vel; park[128] = rp.pos; park[192] = rp.rot;
            const unsigned long long fb = __ballot(fl);
            const bool fell = (half ? (unsigned)(fb >> 32) : (unsigned)fb) != 0u;
            float sk = 0.f;
            if (lb < 4) sk = strict::sum_bodies(park_all + (threadIdx.x >> 6) * LDS_FLOATS_PER_WAVE + lb * 64 + base);
            const float s4[4] = {pull(sk, base), pull(sk, base + 1), pull(sk, base + 2), pull(sk, base + 3)};
            if (lb == 0 && live_env) strict::post_env(Z.b, Z.t, P, mid, t_new, e, s4, fell);
        }
    }
    }
    if constexpr (JOBS) if (!mono) {
        // the step of this pair is complete: jobs of it that have not started yet must not (see the look at the word in the prologue)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            int bid_here = bid;
            asm volatile("" : "+s"(bid_here));
            __hip_atomic_fetch_max(a.job_progress + (bid_here * LL_WPB + (threadIdx.x >> 6)), a.job_epoch * (a.p.nsub + 1) + a.p.nsub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    timeline();
}

// ---- pairing: a wave costs the union of its two envs' contact structure, so envs are handed to waves in descending order of
// their contact load (heavy waves first also keeps the tail of the launch short; the heaviest quarter each next to one of the lightest).
// A counting sort without a sorting kernel: every env draws an arrival index in its load bin at the end of the physics kernel (atomics)
// and appends itself to the bin's arrival list, the workgroup that finishes last scans the 256 bin counts, and the next launch looks
// its envs up (prologue of physics_ll_kernel).  The order inside a bin depends on arrival, which is harmless: an env's arithmetic does
// not depend on the env it shares a wave with (tests: bit-identical results).  The explicit slot -> env table below is built on
// demand only (v2p_env_debug_pairing).
__global__ void pair_scatter_kernel(PairView pv, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) pair_scatter(pv, e);
}

// ---- stand-alone pre-physics (the staged API, v2p_env_pre_physics): one thread per action component
__global__ void env_pre_kernel(PhysArgs a, PairView pv) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t e = tid / NACT;
    const int c = (int)(tid - e * NACT);
    if (e >= a.n) return;
    if (c == 0 && pv.perm) pair_scatter(pv, e);
    const bool dead = a.reset[e] == 1;
    float act = a.actions[tid];
    if (dead) { act = 0.f; a.actions[tid] = 0.f; }  // in place on the caller's tensor, like the reference
    if (c < NDOF) {
        const float tar = strict::pd_clamp(act, a.x_dof[(e * NDOF + c) * 2], a.p.pd_tar_lim);
        a.pd_target[e * NDOF + c] = tar;
        a.ctrl[CIDX(CT_PD + c)] = tar;
    } else if (c == NDOF || c == NDOF + 3) {
        const float a1 = dead ? 0.f : a.actions[tid + 1], a2 = dead ? 0.f : a.actions[tid + 2];
        const strict::V3 w = strict::residual_wrench(a.x_rb + e * NB * 13 + 3, act, a1, a2, c == NDOF ? a.p.res_force_scale : a.p.res_torque_scale);
        const int base = c == NDOF ? CT_FORCE : CT_TORQUE;
        a.ctrl[CIDX(base + 0)] = w.x; a.ctrl[CIDX(base + 1)] = w.y; a.ctrl[CIDX(base + 2)] = w.z;
    }
}

int launch_env_pre(v2p_env* env, float* actions, hipStream_t s) {
    PhysArgs a = {};
    a.ctrl = env->ctrl;
    a.actions = actions;
    a.reset = env->buf.reset;
    a.pd_target = env->buf.pd_target;
    a.x_dof = env->buf.dof_state;
    a.x_rb = env->buf.rb_state;
    a.n = env->n;
    a.p = env->p;
    PairView pv{nullptr, nullptr, nullptr, nullptr, 0, 0};  // (the physics kernel looks its envs up itself: nothing to scatter here)
    const int64_t threads = env->n * NACT;
    hipLaunchKernelGGL(env_pre_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a, pv);
    return check_hip(hipGetLastError(), "env_pre_kernel");
}

bool env_pairing_on(const v2p_env* env) { return env->pair_period > 0 && env->schedule == 0 && env->p.enable_contact && env->n > 2; }

PairView env_pair_view(const v2p_env* env) {
    int mix = (int)(env->n * (int64_t)env->pair_mix_permille / 1000);
    if (2 * mix > env->n) mix = (int)(env->n / 2);
    return PairView{env->pair_key, env->pair_pos, env->pair_start, env->perm, (int32_t)env->n, mix};
}

int launch_env_pairing(v2p_env* env, hipStream_t s) {
    // (v2p_env_debug_pairing only: the wave order the next launch will look up, as an explicit slot -> env table)
    hipLaunchKernelGGL(pair_scatter_kernel, dim3((unsigned)((env->n + 255) / 256)), dim3(256), 0, s, env_pair_view(env), env->n);
    return check_hip(hipGetLastError(), "pair_scatter_kernel");
}

int launch_env_physics_ll(v2p_env* env, hipStream_t s, float* actions, int* fused_post) {
    if (fused_post) *fused_post = 0;
    const bool paired = env_pairing_on(env);
    PhysArgs a = {};
    const int buf = env->pair_buf;
    a.pl_start = (paired && env->pair_have) ? env->pair_starts[buf] : nullptr;
    a.pl_list = env->pair_list[buf];
    a.pl_list_next = env->pair_list[1 - buf];
    a.pl_mix = env_pair_view(env).mix;
    a.pl_slot_env = env->pair_slot_env;
    a.pair_key = env->pair_key;
    a.pair_pos = env->pair_pos;
    a.pair_hist = paired ? env->pair_hist : nullptr;
    a.pair_start = env->pair_starts[1 - buf];
    a.pair_done = env->pair_done;
    a.model = env->model->dev;
    a.state = env->state;
    a.ctrl = env->ctrl;
    a.actions = actions;  // non-null: pre-physics runs in this kernel's prologue
    a.par_pack[0] = a.par_pack[1] = 0ull;
    for (int i = 0; i < NB; ++i) {
        const int par = env->model->host.parents[i] < 0 ? 0 : env->model->host.parents[i];
        a.par_pack[i / 12] |= (unsigned long long)par << (5 * (i % 12));
    }
    a.reset = env->buf.reset;
    a.pd_target = env->buf.pd_target;
    a.out = env->out;
    a.ws = env->ws;
    a.contact_ids = env->contact_ids;
    a.contact_ids_sub = env->contact_ids_sub;
    a.x_root = env->buf.root_states;
    a.x_dof = env->buf.dof_state;
    a.x_rb = env->buf.rb_state;
    a.x_contact = env->buf.contact_force;
    a.x_dof_force = env->buf.dof_force;
    a.prof = env->prof;
    a.prof_heavy = debug_env("V2P_PHASE_HEAVY") ? 1 : 0;  // diagnostics: sample the 8 heaviest waves instead of every 64th
    a.wave_times = env->wave_times;
    a.n = env->n;
    a.p = env->p;
    unsigned blocks = (unsigned)((env->n + 2 * LL_WPB - 1) / (2 * LL_WPB));
    a.shapes = env->shapes_dev;
    a.env_shape = env->env_shape_dev;
    a.shape_aug = env->shape_aug_dev;
    const bool multi = env->num_shapes > 1;  // per-env shapes: hull vertices come from the shape tables instead of the LDS copy
    const dim3 grid(blocks), block(64 * LL_WPB);
    const size_t lds = sizeof(float) * LDS_FLOATS_PER_WAVE * LL_WPB;
    const bool tgs = env->p.solver_type == 1;
#if defined(V2P_LL_TIMELINE)
    const bool diag = a.prof != nullptr;
#else
    const bool diag = a.prof || a.wave_times;
#endif
    if (env->ball) a.ball = *env->ball;
    // ONE instantiation whether the launch is cut into substep jobs or not (job_mono = every pair: whole control steps per workgroup,
    // nothing handed over): "substep jobs are invisible" holds by construction - two instantiations of the template are two
    // compilations, and under -fassociative-math nothing makes them round alike
    a.job_blocks = (int)blocks;
    a.job_progress = env->job_progress;
    a.job_hand = env->job_hand;
    a.job_timeout_spins = env->job_timeout_spins;
    a.job_interleave = env->job_interleave;
    a.job_len = 1;
    a.job_lead = 1;
    a.job_mono = (int)blocks;
    auto job_grid = [&](int& rc) -> dim3 {
        rc = V2P_OK;
        const bool cut = env->substep_jobs && env->job_progress && blocks > 1 && (int)blocks > env->job_min_blocks;
        if (cut) {
            // substep jobs: one launch of job_mono + nsub x (blocks - job_mono) workgroups, substep-major
            a.job_epoch = ++env->job_epoch;
            if (env->job_epoch > (1 << 30) / (env->p.nsub + 1) - 2) env->job_epoch = 0;  // (wraps before the progress words overflow; a wrap needs them cleared)
            if (env->job_epoch == 0) {
                rc = check_hip(hipMemsetAsync(env->job_progress, 0, sizeof(int) * (size_t)job_wave_slots(env->n), s), "hipMemsetAsync(job_progress)");  // (not the error word behind them)
                a.job_epoch = env->job_epoch = 1;
            }
            a.job_mono = (int)(blocks * (unsigned)env->job_mono_permille / 1000u);
        }
        a.job_len = !cut ? 1 : (env->job_len >= 1 ? env->job_len : (((int)blocks >= env->job_len2_blocks && env->p.nsub % 2 == 0) ? 2 : 1));
        // (job_lead: substeps of the first job of a cut pair; 0 / out of range = job_len, i.e. jobs of equal length; -1 = the engine's
        // choice: with one-substep jobs the first job takes two substeps - one hand-over less per pair (a third of the hand-over traffic
        // at four substeps) while the jobs that END a launch stay one substep long; measured +0.3 % at 8192 envs, +1.3 % at 12288, three
        // substeps in the first job -4.7 %: profiles/r04_job_lead.txt)
        // (not with a ball: 12.68 vs 12.79 M)
        const int lead_req = env->job_lead >= 0 ? env->job_lead : ((a.job_len == 1 && env->p.nsub >= 4 && !env->ball) ? 2 : 0);
        a.job_lead = (cut && lead_req >= 1 && lead_req < env->p.nsub) ? lead_req : a.job_len;
        const unsigned njobs = 1u + (unsigned)((env->p.nsub - a.job_lead + a.job_len - 1) / a.job_len);
        return dim3((unsigned)a.job_mono + (blocks - (unsigned)a.job_mono) * njobs);
    };
    // every production instantiation is cut into substep jobs and runs post-physics in the epilogue of an env's last job (v2p_env_step);
    // the instrumented build (DIAG) exists for the headline configuration only and keeps whole control steps per workgroup
    auto with_post = [&]() {
        if (fused_post && actions && env->mlib) {
            a.post.b = env->buf;
            a.post.t = env->mlib->t;
            a.post.motion_id = env->motion_id;
            a.post.cur = env->cur_target;
            a.post.on = 1;
            *fused_post = 1;
        }
    };
    if (env->p.joint_limits && !env->p.enable_contact) {
        set_error("physics: joint limits run with contacts on");
        return V2P_ERR_UNSUPPORTED;
    }
    if (env->ball && !env->p.enable_contact) {
        set_error("physics: racket + ball runs with contacts on");
        return V2P_ERR_UNSUPPORTED;
    }
    if (diag && env->p.enable_contact && !tgs && !multi && !env->ball && !env->p.joint_limits && env->p.friction_frame == 0) {
        hipLaunchKernelGGL((physics_ll_kernel<true, false, false, true, false, false, false>), grid, block, lds, s, a);
    } else {
        int rc0;
        const dim3 jgrid = job_grid(rc0);
        if (rc0 != V2P_OK) return rc0;
        with_post();
        const bool lim = env->p.joint_limits != 0, ball = env->ball != nullptr, con = env->p.enable_contact != 0;
        const bool vfric = env->p.friction_frame == 1 && con;
#define V2P_LL_LAUNCH(C, T, B, L) do { \
            if (vfric && C) { \
                if (multi) hipLaunchKernelGGL((physics_ll_kernel<C, true, T, false, B, true, L, C>), jgrid, block, lds, s, a); \
                else hipLaunchKernelGGL((physics_ll_kernel<C, false, T, false, B, true, L, C>), jgrid, block, lds, s, a); \
            } else if (multi) hipLaunchKernelGGL((physics_ll_kernel<C, true, T, false, B, true, L>), jgrid, block, lds, s, a); \
            else hipLaunchKernelGGL((physics_ll_kernel<C, false, T, false, B, true, L>), jgrid, block, lds, s, a); } while (0)
        if (lim && ball && tgs) V2P_LL_LAUNCH(true, true, true, true);
        else if (lim && ball) V2P_LL_LAUNCH(true, false, true, true);
        else if (lim && tgs) V2P_LL_LAUNCH(true, true, false, true);
        else if (lim) V2P_LL_LAUNCH(true, false, false, true);
        else if (ball && tgs) V2P_LL_LAUNCH(true, true, true, false);
        else if (ball) V2P_LL_LAUNCH(true, false, true, false);
        else if (con && tgs) V2P_LL_LAUNCH(true, true, false, false);
        else if (con) V2P_LL_LAUNCH(true, false, false, false);
        else V2P_LL_LAUNCH(false, false, false, false);
#undef V2P_LL_LAUNCH
    }
    if (paired) {  // the tables this launch has filled are what the next one reads
        env->pair_buf = 1 - buf;
        env->pair_start = env->pair_starts[env->pair_buf];
        env->pair_have = 1;
    }
    return check_hip(hipGetLastError(), "physics_ll_kernel");
}

}  // namespace v2p
