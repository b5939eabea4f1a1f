// Device-side fp32 vector / quaternion helpers for gfx950 (one value per lane, no arrays
// that would force scratch).  Quaternions are xyzw.  The task-side helpers restate
// embodied_pose/utils/torch_utils.py with the same thresholds so results match the
// reference within float32 rounding.
#pragma once
#include <hip/hip_runtime.h>

namespace v2p {

struct V3 {
    float x, y, z;
};
struct Q4 {
    float x, y, z, w;
};

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
    return Q4{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ Q4 qconj(Q4 q) { return Q4{-q.x, -q.y, -q.z, q.w}; }
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
    float n = rsqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return Q4{q.x * n, q.y * n, q.z * n, q.w * n};
}

// 3x3 matrix, row-major
struct M3 {
    float m[9];
};
__device__ __forceinline__ M3 q2mat(Q4 q) {
    M3 R;
    float x = q.x, y = q.y, z = q.z, w = q.w;
    R.m[0] = 1.f - 2.f * (y * y + z * z); R.m[1] = 2.f * (x * y - z * w);       R.m[2] = 2.f * (x * z + y * w);
    R.m[3] = 2.f * (x * y + z * w);       R.m[4] = 1.f - 2.f * (x * x + z * z); R.m[5] = 2.f * (y * z - x * w);
    R.m[6] = 2.f * (x * z - y * w);       R.m[7] = 2.f * (y * z + x * w);       R.m[8] = 1.f - 2.f * (x * x + y * y);
    return R;
}
__device__ __forceinline__ V3 mul(const M3& R, V3 v) {
    return V3{R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
              R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
__device__ __forceinline__ V3 mulT(const M3& R, V3 v) {
    return V3{R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z,
              R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z};
}

// ---- reference task math (embodied_pose/utils/torch_utils.py) ---------------------------
// my_quat_rotate (:70-79)
__device__ __forceinline__ V3 ref_quat_rotate(Q4 q, V3 v) {
    V3 qv{q.x, q.y, q.z};
    float s = 2.f * q.w * q.w - 1.f;
    V3 a = s * v;
    V3 b = (cross(qv, v) * q.w) * 2.f;
    V3 c = (qv * dot(qv, v)) * 2.f;
    return a + b + c;
}
// normalize_angle = atan2(sin, cos)
__device__ __forceinline__ float ref_normalize_angle(float a) { return atan2f(sinf(a), cosf(a)); }
// quat_to_angle_axis (:82-102): returns angle, writes axis
__device__ __forceinline__ float ref_quat_to_angle_axis(Q4 q, V3& axis) {
    float s2 = 1.f - q.w * q.w;
    float sin_theta = sqrtf(s2);  // nan for |w|>1, which fails the mask below exactly like torch
    float angle = ref_normalize_angle(2.f * acosf(q.w));
    bool ok = fabsf(sin_theta) > 1e-5f;
    float inv = 1.f / sin_theta;
    axis = ok ? V3{q.x * inv, q.y * inv, q.z * inv} : V3{0.f, 0.f, 1.f};
    return ok ? angle : 0.f;
}
// quat_to_exp_map (:113-119)
__device__ __forceinline__ V3 ref_quat_to_exp_map(Q4 q) {
    V3 ax;
    float ang = ref_quat_to_angle_axis(q, ax);
    return ang * ax;
}
// quat_from_angle_axis (isaacgym.torch_utils): axis normalised, result re-normalised
__device__ __forceinline__ Q4 ref_quat_from_angle_axis(float angle, V3 axis) {
    float n = fmaxf(sqrtf(dot(axis, axis)), 1e-9f);
    float s, c;
    sincosf(0.5f * angle, &s, &c);
    Q4 q{axis.x / n * s, axis.y / n * s, axis.z / n * s, c};
    float qn = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-9f);
    return Q4{q.x / qn, q.y / qn, q.z / qn, q.w / qn};
}
// exp_map_to_quat (:144-166)
__device__ __forceinline__ Q4 ref_exp_map_to_quat(V3 e) {
    float ang = sqrtf(dot(e, e));
    float angn = ref_normalize_angle(ang);
    bool ok = fabsf(angn) > 1e-5f;
    float inv = 1.f / ang;
    V3 ax = ok ? V3{e.x * inv, e.y * inv, e.z * inv} : V3{0.f, 0.f, 1.f};
    return ref_quat_from_angle_axis(ok ? angn : 0.f, ax);
}
// slerp (:169-190)
__device__ __forceinline__ Q4 ref_slerp(Q4 q0, Q4 q1, float t) {
    float c = q0.x * q1.x + q0.y * q1.y + q0.z * q1.z + q0.w * q1.w;
    if (c < 0.f) q1 = Q4{-q1.x, -q1.y, -q1.z, -q1.w};
    c = fabsf(c);
    float half = acosf(c);
    float sh = sqrtf(1.f - c * c);
    float ra = sinf((1.f - t) * half) / sh;
    float rb = sinf(t * half) / sh;
    Q4 r{ra * q0.x + rb * q1.x, ra * q0.y + rb * q1.y, ra * q0.z + rb * q1.z, ra * q0.w + rb * q1.w};
    if (fabsf(sh) < 0.001f) r = Q4{0.5f * q0.x + 0.5f * q1.x, 0.5f * q0.y + 0.5f * q1.y, 0.5f * q0.z + 0.5f * q1.z, 0.5f * q0.w + 0.5f * q1.w};
    if (fabsf(c) >= 1.f) r = q0;
    return r;
}
// quat_to_tan_norm (:122-134): rotated x axis (tan) and rotated z axis (norm)
__device__ __forceinline__ void ref_quat_to_tan_norm(Q4 q, V3& tan, V3& nrm) {
    tan = ref_quat_rotate(q, V3{1.f, 0.f, 0.f});
    nrm = ref_quat_rotate(q, V3{0.f, 0.f, 1.f});
}
// remove_base_rot (humanoid_smpl_im.py:766-770): q * conj([.5,.5,.5,.5])
__device__ __forceinline__ Q4 ref_remove_base_rot(Q4 q) { return qmul(q, Q4{-0.5f, -0.5f, -0.5f, 0.5f}); }
// calc_heading (:193-204)
__device__ __forceinline__ float ref_calc_heading(Q4 q) {
    V3 d = ref_quat_rotate(q, V3{1.f, 0.f, 0.f});
    return atan2f(d.y, d.x);
}
// calc_heading_quat (:206-217) / _inv (:219-243)
__device__ __forceinline__ Q4 ref_heading_quat(float heading) { return ref_quat_from_angle_axis(heading, V3{0.f, 0.f, 1.f}); }

// ---- well-conditioned variants used inside the physics step ------------------------------
// exponential map of a unit quaternion with the reference's sign convention (angle wrapped
// to (-pi, pi]); atan2 form instead of acos(w) so small joint angles keep full precision.
__device__ __forceinline__ V3 quat_to_expmap_stable(Q4 q) {
    float s = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z);
    float ang = 2.f * atan2f(s, q.w);
    if (ang > 3.14159265358979f) ang -= 6.28318530717959f;
    float k = s > 1e-5f ? ang / s : 0.f;
    return V3{q.x * k, q.y * k, q.z * k};
}
// rotation vector -> quaternion
__device__ __forceinline__ Q4 rotvec_to_quat(V3 v) {
    float a2 = dot(v, v);
    float a = sqrtf(a2);
    float s, c;
    sincosf(0.5f * a, &s, &c);
    float k = a > 1e-6f ? s / a : 0.5f - a2 * (1.f / 48.f);
    return Q4{v.x * k, v.y * k, v.z * k, c};
}

}  // namespace v2p
