// Distance of a point from the convex hull of a vertex list: GJK on the points v_i - c with the closest-point-on-simplex rules of
// Ericson ("Real-Time Collision Detection", 5.1).  One lane = one hull (the ball x link contacts of physics_ll.hip); everything lives
// in registers: the simplex is kept compact with compile-time indices only (no dynamically indexed arrays, which would go to scratch).
// The model it serves is stated in oracle/phys/v2p_phys_oracle.c (`hull_closest`).
#pragma once
#include "phys_math.hpp"

namespace v2p {

// barycentric weights of the point of the segment / triangle closest to the origin
__device__ __forceinline__ void gjk_seg(V3 a, V3 b, float& wa, float& wb) {
    const V3 ab = b - a;
    const float den = dot(ab, ab);
    float t = den > 0.f ? -dot(a, ab) / den : 0.f;
    t = fminf(fmaxf(t, 0.f), 1.f);
    wa = 1.f - t;
    wb = t;
}
__device__ __forceinline__ void gjk_tri(V3 a, V3 b, V3 c, float& wa, float& wb, float& wc) {
    const V3 ab = b - a, ac = c - a;
    const float d1 = -dot(ab, a), d2 = -dot(ac, a);
    wa = wb = wc = 0.f;
    if (d1 <= 0.f && d2 <= 0.f) { wa = 1.f; return; }
    const float d3 = -dot(ab, b), d4 = -dot(ac, b);
    if (d3 >= 0.f && d4 <= d3) { wb = 1.f; return; }
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { const float v = d1 / (d1 - d3); wa = 1.f - v; wb = v; return; }
    const float d5 = -dot(ab, c), d6 = -dot(ac, c);
    if (d6 >= 0.f && d5 <= d6) { wc = 1.f; return; }
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { const float u = d2 / (d2 - d6); wa = 1.f - u; wc = u; return; }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && d4 - d3 >= 0.f && d5 - d6 >= 0.f) { const float u = (d4 - d3) / ((d4 - d3) + (d5 - d6)); wb = 1.f - u; wc = u; return; }
    const float den = 1.f / (va + vb + vc);
    wb = vb * den;
    wc = vc * den;
    wa = 1.f - wb - wc;
}
// one face (a b c) of a tetrahedron whose fourth vertex is d: when the origin lies beyond it, the closest point of the face competes
__device__ __forceinline__ void gjk_face(V3 a, V3 b, V3 c, V3 d, float& best, bool& any, float& wa, float& wb, float& wc, float& wd) {
    const V3 n = cross(b - a, c - a);
    const float so = -dot(n, a), sd = dot(n, d - a);
    if (so * sd < 0.f || sd == 0.f) {
        float ta, tb, tc;
        gjk_tri(a, b, c, ta, tb, tc);
        const V3 q = ta * a + tb * b + tc * c;
        const float dd = dot(q, q);
        if (dd < best) { best = dd; any = true; wa = ta; wb = tb; wc = tc; wd = 0.f; }
    }
}

// verts: the hull's vertices (body frame), c: the point.  Returns the distance; p = the closest point of the hull (c itself, distance
// 0, when c lies inside).
template <typename VertexOf>
__device__ __forceinline__ float hull_closest(VertexOf vertex, int nv, V3 c, V3& p) {
    V3 s0 = vertex(0) - c, s1{0.f, 0.f, 0.f}, s2{0.f, 0.f, 0.f}, s3{0.f, 0.f, 0.f};
    int i0 = 0, i1 = -1, i2 = -1, i3 = -1, n = 1;
    V3 v = s0;
    for (int it = 0; it < 32; ++it) {
        int best = 0;
        float bd = 3.0e38f;
        for (int k = 0; k < nv; ++k) {
            const float d = dot(v, vertex(k) - c);
            if (d < bd) { bd = d; best = k; }
        }
        const float vv = dot(v, v);
        if (vv - bd <= 1e-6f * vv + 1e-14f) break;  // no vertex lies closer along -v: v is the closest point
        if (best == i0 || (n > 1 && best == i1) || (n > 2 && best == i2) || (n > 3 && best == i3)) break;
        const V3 w = vertex(best) - c;
        if (n == 1) { s1 = w; i1 = best; } else if (n == 2) { s2 = w; i2 = best; } else { s3 = w; i3 = best; }
        ++n;
        float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
        bool inside = false;
        if (n == 2) gjk_seg(s0, s1, w0, w1);
        else if (n == 3) gjk_tri(s0, s1, s2, w0, w1, w2);
        else {
            float bestd = 3.0e38f;
            bool any = false;
            gjk_face(s0, s1, s2, s3, bestd, any, w0, w1, w2, w3);
            gjk_face(s0, s2, s3, s1, bestd, any, w0, w2, w3, w1);
            gjk_face(s0, s3, s1, s2, bestd, any, w0, w3, w1, w2);
            gjk_face(s1, s3, s2, s0, bestd, any, w1, w3, w2, w0);
            inside = !any;
        }
        if (inside) { v = V3{0.f, 0.f, 0.f}; break; }
        v = w0 * s0 + w1 * s1 + w2 * s2 + w3 * s3;
        // keep the vertices that carry the closest point, in order (compile-time indices only)
        V3 t0 = s0, t1 = s1, t2 = s2, t3 = s3;
        int j0 = i0, j1 = i1, j2 = i2, j3 = i3, m = 0;
        auto keep = [&](V3 sv, int si) {
            if (m == 0) { t0 = sv; j0 = si; } else if (m == 1) { t1 = sv; j1 = si; } else if (m == 2) { t2 = sv; j2 = si; } else { t3 = sv; j3 = si; }
            ++m;
        };
        if (w0 > 0.f) keep(s0, i0);
        if (n > 1 && w1 > 0.f) keep(s1, i1);
        if (n > 2 && w2 > 0.f) keep(s2, i2);
        if (n > 3 && w3 > 0.f) keep(s3, i3);
        s0 = t0; s1 = t1; s2 = t2; s3 = t3;
        i0 = j0; i1 = j1; i2 = j2; i3 = j3;
        n = m;
        if (n == 0 || dot(v, v) < 1e-14f) { v = V3{0.f, 0.f, 0.f}; break; }
    }
    p = c + v;
    return sqrtf(dot(v, v));
}

}  // namespace v2p
