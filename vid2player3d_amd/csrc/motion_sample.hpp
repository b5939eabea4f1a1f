// Reference-motion sampler, device side.  Restates MotionLib.get_motion_state
// (embodied_pose/utils/motion_lib.py:164-266) + _calc_frame_blend (:427-436) +
// _local_rotation_to_dof (:460-488) for one (query, body) pair per thread, so that the two
// frame rows of a query (24 bodies x {3,4,4} floats, contiguous in the reference's [F,24,*]
// tables) are read by 24 adjacent lanes.
#pragma once
#include "v2p_internal.hpp"
#include "v2p_math.hpp"

namespace v2p {

struct MSOut {
    float* p[9];       // root_pos root_rot dof_pos root_vel root_ang_vel dof_vel key_pos rb_pos rb_rot (NULL = skip)
    int64_t stride[9]; // row stride of each output, in floats
};

struct FrameRef {
    int64_t f0l, f1l;
    float blend;
    float min_vh;  // height adjustment (0 when adjust_height is off)
};

__device__ __forceinline__ FrameRef frame_lookup(const v2p_motion_tables& t, int64_t id, float time, int adjust_height, float ground_tol) {
    float len = t.motion_lengths[id];
    int64_t nf = t.motion_num_frames[id];
    float dt = t.motion_dt[id];
    float phase = time / len;
    phase = fminf(fmaxf(phase, 0.f), 1.f);
    int64_t f0 = (int64_t)(phase * (float)(nf - 1));
    int64_t f1 = f0 + 1 < nf - 1 ? f0 + 1 : nf - 1;
    FrameRef r;
    r.blend = (time - (float)f0 * dt) / dt;  // not clamped: exceeds 1 past the clip end
    int64_t start = t.length_starts[id];
    r.f0l = f0 + start;
    r.f1l = f1 + start;
    r.min_vh = adjust_height ? t.motion_min_verts_h[id] - ground_tol : 0.f;
    return r;
}

__device__ __forceinline__ Q4 load_q4(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    return Q4{v.x, v.y, v.z, v.w};
}

// what body j of one query contributes
struct BodySample {
    V3 pos;      // blended global translation (height-adjusted)
    Q4 rot;      // slerped global rotation
    V3 dof_pos;  // exp-map of the slerped local rotation (j >= 1)
    V3 dof_vel;  // frame-0 dof velocity (j >= 1)
    V3 root_vel, root_ang_vel;  // frame-0 root velocities (j == 0)
};

__device__ __forceinline__ BodySample sample_body_values(const v2p_motion_tables& t, const FrameRef& fr, int j) {
    BodySample r;
    const float b = fr.blend;
    const float a = 1.f - b;
    const float* g0 = t.gts + (fr.f0l * NB + j) * 3;
    const float* g1 = t.gts + (fr.f1l * NB + j) * 3;
    r.pos = V3{a * g0[0] + b * g1[0], a * g0[1] + b * g1[1], a * g0[2] + b * g1[2]};
    r.pos.z -= fr.min_vh;
    r.rot = ref_slerp(load_q4(t.grs + (fr.f0l * NB + j) * 4), load_q4(t.grs + (fr.f1l * NB + j) * 4), b);
    r.dof_pos = r.dof_vel = r.root_vel = r.root_ang_vel = V3{0.f, 0.f, 0.f};
    if (j == 0) {
        const float* s = t.grvs + fr.f0l * 3;
        r.root_vel = V3{s[0], s[1], s[2]};
        s = t.gravs + fr.f0l * 3;
        r.root_ang_vel = V3{s[0], s[1], s[2]};
    } else {
        Q4 lr = ref_slerp(load_q4(t.lrs + (fr.f0l * NB + j) * 4), load_q4(t.lrs + (fr.f1l * NB + j) * 4), b);
        r.dof_pos = ref_quat_to_exp_map(lr);
        const float* s = t.dvs + fr.f0l * NDOF + 3 * (j - 1);  // velocities come from frame 0 only
        r.dof_vel = V3{s[0], s[1], s[2]};
    }
    return r;
}

__device__ __forceinline__ void st3(float* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
__device__ __forceinline__ void st4(float* d, Q4 q) { d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w; }

__device__ __forceinline__ void store_body_sample(const v2p_motion_tables& t, const BodySample& r, int j, int64_t row, const MSOut& o) {
    if (o.p[7]) st3(o.p[7] + row * o.stride[7] + j * 3, r.pos);
    if (o.p[8]) st4(o.p[8] + row * o.stride[8] + j * 4, r.rot);
    if (j == 0) {
        if (o.p[0]) st3(o.p[0] + row * o.stride[0], r.pos);
        if (o.p[1]) st4(o.p[1] + row * o.stride[1], r.rot);
        if (o.p[3]) st3(o.p[3] + row * o.stride[3], r.root_vel);
        if (o.p[4]) st3(o.p[4] + row * o.stride[4], r.root_ang_vel);
    } else {
        if (o.p[2]) st3(o.p[2] + row * o.stride[2] + 3 * (j - 1), r.dof_pos);
        if (o.p[5]) st3(o.p[5] + row * o.stride[5] + 3 * (j - 1), r.dof_vel);
    }
    if (o.p[6]) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (t.key_body_ids[k] == j) st3(o.p[6] + row * o.stride[6] + 3 * k, r.pos);
    }
}

__device__ __forceinline__ void sample_body(const v2p_motion_tables& t, const FrameRef& fr, int j, int64_t row, const MSOut& o) {
    BodySample r = sample_body_values(t, fr, j);
    store_body_sample(t, r, j, row, o);
}

// outputs packed into one [rows, 331] block
__device__ __forceinline__ MSOut packed_out(float* base) {
    MSOut o;
    const int off[9] = {MS_ROOT_POS, MS_ROOT_ROT, MS_DOF_POS, MS_ROOT_VEL, MS_ROOT_ANG_VEL, MS_DOF_VEL, MS_KEY_POS, MS_RB_POS, MS_RB_ROT};
#pragma unroll
    for (int i = 0; i < 9; ++i) { o.p[i] = base + off[i]; o.stride[i] = MSD; }
    return o;
}

}  // namespace v2p
