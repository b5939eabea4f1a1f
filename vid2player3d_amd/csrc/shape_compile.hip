// Per-clip body assets on the device ("GPU-side model blobs for thousands of body shapes").
//
// The reference builds one humanoid asset per sampled clip (humanoid_smpl_im.py:255-296): the SMPL vertices of every body become a
// convex hull (uhc/smpllib/smpl_local_robot.py:79-143: scipy ConvexHull -> STL -> decimation), Isaac Gym cooks the hull and
// integrates its mass properties at the geom density.  AMASS means thousands of clips = thousands of (shape, body) hulls at task
// construction; vid2player3d_amd/body_shapes.py does one in ~12 ms of numpy.  This kernel does the same computation -
//
//     cloud [n,3] -> convex hull (incremental insertion, farthest point first) -> at most `max_verts` support vertices
//                 -> hull of those -> mass, centre of mass, inertia about it (signed tetrahedra over the hull's faces)
//
// - with ONE WAVEFRONT PER (shape, body) JOB: lanes are the remaining points while distances to the face planes are evaluated (planes
// broadcast from LDS), lanes are the faces while visibility, the horizon and the mass integrals are evaluated; face lists, planes and
// the remaining-point list live in LDS (one workgroup = one wave, up to ~150 KB of the CU's 160 KB for clouds of ~1500 points), all
// arithmetic in float64 (init-time path: exactness over speed; MI355X's vector float64 rate is not the limit here).  The algorithm, its
// tie-breaking and its epsilon rule are body_shapes.convex_hull / reduce_hull / hull_mass_properties_faces, which stay the CPU checker.
#include <math.h>

#include "v2p_dev.hpp"

namespace v2p {

namespace {

constexpr int SC_WAVE = 64;

struct ShapeCompileArgs {
    const double* pts;         // [total points][3]
    const int32_t* job_off;    // [jobs + 1] first point of every job
    const double* dirs;        // Fibonacci direction tables, concatenated [..][3]
    const int32_t* dir_off;    // [num_tables + 1]
    int32_t num_tables;
    int32_t jobs;
    int32_t max_pts;           // largest cloud (sizes the LDS arrays)
    int32_t max_verts;
    double density;
    double eps_rel;
    double* mass;              // [jobs]
    double* com;               // [jobs][3]
    double* inertia;           // [jobs][9]
    int32_t* num_verts;        // [jobs]
    int32_t* vert_ids;         // [jobs][max_verts] indices into the job's cloud (ascending)
    double* verts;             // [jobs][max_verts][3]
    int32_t* status;           // [jobs] 0 ok, 1 fewer than 4 points, 2 coplanar, 3 face capacity, 4 reduction failed
};

struct Face { unsigned short a, b, c, pad; };

// the LDS arrays of one wave, carved out of the dynamic allocation
struct Lds {
    double4* plane;        // [maxf] nx ny nz D (after the first hull: coordinates of the hull vertices, 3 doubles each)
    Face* face;            // [maxf]
    unsigned* hor;         // [maxf] horizon edges a << 16 | b
    unsigned short* vlist; // [maxf] visible faces
    unsigned short* rem;   // [maxn] remaining points (then: hull vertex list)
    unsigned char* mark;   // [maxn]
};

__device__ inline double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, SC_WAVE);
    return v;
}

__device__ inline int wave_sum_i(int v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, SC_WAVE);
    return v;
}

// (value, index): the larger value, on ties the smaller index (numpy argmax takes the first maximum)
__device__ inline void wave_argmax(double& v, int& i) {
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_xor(v, o, SC_WAVE);
        int oi = __shfl_xor(i, o, SC_WAVE);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

__device__ inline void wave_argmin(double& v, int& i) {
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_xor(v, o, SC_WAVE);
        int oi = __shfl_xor(i, o, SC_WAVE);
        if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

struct D3 { double x, y, z; };
__device__ inline D3 ldp(const double* P, int i) { return D3{P[3 * i], P[3 * i + 1], P[3 * i + 2]}; }
__device__ inline D3 sub3(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline D3 cross3(D3 u, D3 v) { return D3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x}; }
__device__ inline double dot3(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// unit normal and offset of the plane through the face (body_shapes.planes_of)
__device__ inline double4 plane_of(const double* P, Face f) {
    D3 a = ldp(P, f.a);
    D3 n = cross3(sub3(ldp(P, f.b), a), sub3(ldp(P, f.c), a));
    double inv = 1.0 / sqrt(dot3(n, n));
    n.x *= inv; n.y *= inv; n.z *= inv;
    return double4{n.x, n.y, n.z, dot3(n, a)};
}

// Convex hull of P[0..n) (body_shapes.convex_hull): faces wound outward in L.face / L.plane; returns the face count, < 0 = -status.
// One wave; every lane must call it.
__device__ int hull(const double* P, int n, double eps_rel, int maxf, Lds L) {
    const int lane = threadIdx.x;
    if (n < 4) return -1;
    // scale = max |p - mean|, the initial tetrahedron: extreme pair, farthest from their line, farthest from their plane
    double sx = 0, sy = 0, sz = 0;
    for (int i = lane; i < n; i += SC_WAVE) { D3 p = ldp(P, i); sx += p.x; sy += p.y; sz += p.z; }
    sx = wave_sum(sx) / n; sy = wave_sum(sy) / n; sz = wave_sum(sz) / n;
    double sc = 0;
    {
        double bx = 1e300; int bi = 0x7fffffff;
        for (int i = lane; i < n; i += SC_WAVE) {
            D3 p = ldp(P, i);
            sc = fmax(sc, fmax(fabs(p.x - sx), fmax(fabs(p.y - sy), fabs(p.z - sz))));
            if (p.x < bx) { bx = p.x; bi = i; }
        }
        for (int o = 32; o > 0; o >>= 1) sc = fmax(sc, __shfl_xor(sc, o, SC_WAVE));
        wave_argmin(bx, bi);
        sx = bx; sy = (double)bi;  // (reuse: i0 below)
    }
    const double eps = eps_rel * (sc + 1e-300);
    const int i0 = (int)sy;
    const D3 p0 = ldp(P, i0);
    int i1, i2, i3;
    {
        double bv = -1; int bi = 0x7fffffff;
        for (int i = lane; i < n; i += SC_WAVE) { D3 d = sub3(ldp(P, i), p0); double v = sqrt(dot3(d, d)); if (v > bv) { bv = v; bi = i; } }
        wave_argmax(bv, bi);
        i1 = bi;
    }
    const D3 d01 = sub3(ldp(P, i1), p0);
    {
        double bv = -1; int bi = 0x7fffffff;
        for (int i = lane; i < n; i += SC_WAVE) { D3 c = cross3(sub3(ldp(P, i), p0), d01); double v = sqrt(dot3(c, c)); if (v > bv) { bv = v; bi = i; } }
        wave_argmax(bv, bi);
        i2 = bi;
    }
    const D3 nrm = cross3(d01, sub3(ldp(P, i2), p0));
    double d3v;
    {
        double bv = -1; int bi = 0x7fffffff;
        for (int i = lane; i < n; i += SC_WAVE) { double v = fabs(dot3(sub3(ldp(P, i), p0), nrm)); if (v > bv) { bv = v; bi = i; } }
        wave_argmax(bv, bi);
        i3 = bi;
        d3v = dot3(sub3(ldp(P, i3), p0), nrm);
    }
    if (fabs(d3v) <= eps * sqrt(dot3(nrm, nrm))) return -2;
    if (d3v > 0) { int t = i1; i1 = i2; i2 = t; }  // (i0, i1, i2) faces away from i3
    int F = 4;
    if (lane < 4) {
        const int fa[4] = {i0, i0, i1, i2}, fb[4] = {i1, i3, i3, i3}, fc[4] = {i2, i1, i2, i0};
        Face f{(unsigned short)fa[lane], (unsigned short)fb[lane], (unsigned short)fc[lane], 0};
        L.face[lane] = f;
        L.plane[lane] = plane_of(P, f);
    }
    int R = 0;
    for (int base = 0; base < n; base += SC_WAVE) {   // remaining = every point but the four, ascending
        int i = base + lane;
        bool keep = i < n && i != i0 && i != i1 && i != i2 && i != i3;
        unsigned long long m = __ballot(keep);
        if (keep) L.rem[R + __popcll(m & ((1ull << lane) - 1))] = (unsigned short)i;
        R += __popcll(m);
    }
    __syncthreads();
    int inserted = -1;  // the point that went in last leaves the list (np.delete): its distance to the faces it has just spawned is zero
                        // only up to the conditioning of their planes - a sliver face can put it "outside" again, forever
    while (R > 0) {
        // signed distance of every remaining point to every face; points outside no face are inside the hull for good
        double best = -1e300; int besti = 0x7fffffff;
        int R2 = 0;
        for (int base = 0; base < R; base += SC_WAVE) {
            int k = base + lane;
            int pi = k < R ? L.rem[k] : 0;
            D3 p = ldp(P, pi);
            double far = -1e300;
            for (int f = 0; f < F; ++f) {
                double4 pl = L.plane[f];
                far = fmax(far, p.x * pl.x + p.y * pl.y + p.z * pl.z - pl.w);
            }
            bool keep = k < R && far > eps && pi != inserted;
            if (keep && far > best) { best = far; besti = pi; }
            unsigned long long m = __ballot(keep);
            __syncthreads();
            if (keep) L.rem[R2 + __popcll(m & ((1ull << lane) - 1))] = (unsigned short)pi;
            R2 += __popcll(m);
        }
        __syncthreads();
        R = R2;
        if (R == 0) break;
        wave_argmax(best, besti);  // the point farthest outside goes in (keeps the face count small)
        const int ip = besti;
        inserted = ip;
        const D3 p = ldp(P, ip);
        // visible faces of ip
        int V = 0;
        for (int base = 0; base < F; base += SC_WAVE) {
            int f = base + lane;
            bool vis = false;
            if (f < F) { double4 pl = L.plane[f]; vis = p.x * pl.x + p.y * pl.y + p.z * pl.z - pl.w > eps; }
            unsigned long long m = __ballot(vis);
            if (vis) L.vlist[V + __popcll(m & ((1ull << lane) - 1))] = (unsigned short)f;
            V += __popcll(m);
        }
        __syncthreads();
        // horizon: directed edges of visible faces whose reverse edge belongs to no visible face
        int H = 0;
        for (int base = 0; base < 3 * V; base += SC_WAVE) {
            int e = base + lane;
            bool hz = false;
            unsigned ea = 0, eb = 0;
            if (e < 3 * V) {
                Face f = L.face[L.vlist[e / 3]];
                int k = e % 3;
                ea = k == 0 ? f.a : (k == 1 ? f.b : f.c);
                eb = k == 0 ? f.b : (k == 1 ? f.c : f.a);
                hz = true;
                for (int j = 0; j < V; ++j) {
                    Face g = L.face[L.vlist[j]];
                    if ((g.a == eb && g.b == ea) || (g.b == eb && g.c == ea) || (g.c == eb && g.a == ea)) { hz = false; break; }
                }
            }
            unsigned long long m = __ballot(hz);
            if (hz) L.hor[H + __popcll(m & ((1ull << lane) - 1))] = (ea << 16) | eb;
            H += __popcll(m);
        }
        if (F - V + H > maxf) return -3;
        __syncthreads();
        // drop the visible faces (order kept), append one face per horizon edge
        for (int j = lane; j < V; j += SC_WAVE) L.face[L.vlist[j]].pad = 1;
        __syncthreads();
        int F2 = 0;
        for (int base = 0; base < F; base += SC_WAVE) {
            int f = base + lane;
            Face fc{0, 0, 0, 1};
            double4 pl{0, 0, 0, 0};
            if (f < F) { fc = L.face[f]; pl = L.plane[f]; }
            bool keep = f < F && fc.pad == 0;
            unsigned long long m = __ballot(keep);
            __syncthreads();
            if (keep) { int d = F2 + __popcll(m & ((1ull << lane) - 1)); L.face[d] = fc; L.plane[d] = pl; }
            F2 += __popcll(m);
        }
        __syncthreads();
        for (int j = lane; j < H; j += SC_WAVE) {
            unsigned e = L.hor[j];
            Face f{(unsigned short)(e >> 16), (unsigned short)(e & 0xffff), (unsigned short)ip, 0};
            L.face[F2 + j] = f;
            L.plane[F2 + j] = plane_of(P, f);
        }
        F = F2 + H;
        __syncthreads();
    }
    return F;
}

__global__ __launch_bounds__(SC_WAVE) void shape_compile_kernel(ShapeCompileArgs a) {
    extern __shared__ __align__(32) unsigned char smem[];
    const int lane = threadIdx.x;
    const int maxn = a.max_pts > a.max_verts ? a.max_pts : a.max_verts;
    const int maxf = 2 * maxn;
    Lds L;
    {
        unsigned char* q = smem;
        L.plane = (double4*)q; q += sizeof(double4) * maxf;
        L.face = (Face*)q; q += sizeof(Face) * maxf;
        L.hor = (unsigned*)q; q += sizeof(unsigned) * maxf;
        L.vlist = (unsigned short*)q; q += sizeof(unsigned short) * maxf;
        L.rem = (unsigned short*)q; q += sizeof(unsigned short) * ((maxn + 3) & ~3);
        L.mark = q;
    }
    for (int job = blockIdx.x; job < a.jobs; job += gridDim.x) {
        const int off = a.job_off[job], n = a.job_off[job + 1] - off;
        const double* P = a.pts + 3 * (size_t)off;
        double* outv = a.verts + 3 * (size_t)job * a.max_verts;
        int32_t* outi = a.vert_ids + (size_t)job * a.max_verts;
        __syncthreads();
        if (n < 0 || n > a.max_pts) {  // an offsets array that disagrees with max_points: the LDS arrays are sized from max_points (advisor r5)
            if (lane == 0) { a.status[job] = 7; a.num_verts[job] = 0; a.mass[job] = 0; }
            continue;
        }
        int F = hull(P, n, a.eps_rel, maxf, L);
        if (F < 0) {
            if (lane == 0) { a.status[job] = -F; a.num_verts[job] = 0; a.mass[job] = 0; }
            continue;
        }
        // hull vertices, ascending (np.unique(faces))
        for (int i = lane; i < n; i += SC_WAVE) L.mark[i] = 0;
        __syncthreads();
        for (int f = lane; f < F; f += SC_WAVE) { Face fc = L.face[f]; L.mark[fc.a] = 1; L.mark[fc.b] = 1; L.mark[fc.c] = 1; }
        __syncthreads();
        int M = 0;
        for (int base = 0; base < n; base += SC_WAVE) {
            int i = base + lane;
            bool hv = i < n && L.mark[i];
            unsigned long long m = __ballot(hv);
            if (hv) L.rem[M + __popcll(m & ((1ull << lane) - 1))] = (unsigned short)i;
            M += __popcll(m);
        }
        __syncthreads();
        int K = M;  // kept vertices
        if (M > a.max_verts) {
            // body_shapes.reduce_hull: the support points of a sphere of directions, fewer directions until at most max_verts distinct ones
            double cx = 0, cy = 0, cz = 0;
            for (int j = lane; j < M; j += SC_WAVE) { D3 p = ldp(P, L.rem[j]); cx += p.x; cy += p.y; cz += p.z; }
            cx = wave_sum(cx) / M; cy = wave_sum(cy) / M; cz = wave_sum(cz) / M;
            double* C = (double*)L.plane;  // centred coordinates of the hull vertices (the planes are dead)
            for (int j = lane; j < M; j += SC_WAVE) { D3 p = ldp(P, L.rem[j]); C[3 * j] = p.x - cx; C[3 * j + 1] = p.y - cy; C[3 * j + 2] = p.z - cz; }
            __syncthreads();
            K = -1;
            for (int t = 0; t < a.num_tables && K < 0; ++t) {
                const int d0 = a.dir_off[t], nd = a.dir_off[t + 1] - d0;
                for (int j = lane; j < M; j += SC_WAVE) L.mark[j] = 0;
                __syncthreads();
                for (int base = 0; base < nd; base += SC_WAVE) {
                    int d = base + lane;
                    if (d < nd) {
                        const double dx = a.dirs[3 * (d0 + d)], dy = a.dirs[3 * (d0 + d) + 1], dz = a.dirs[3 * (d0 + d) + 2];
                        double bv = -1e300; int bj = 0;
                        for (int j = 0; j < M; ++j) {
                            double v = C[3 * j] * dx + C[3 * j + 1] * dy + C[3 * j + 2] * dz;
                            if (v > bv) { bv = v; bj = j; }
                        }
                        L.mark[bj] = 1;
                    }
                }
                __syncthreads();
                int cnt = 0;
                for (int j = lane; j < M; j += SC_WAVE) cnt += L.mark[j];
                cnt = wave_sum_i(cnt);
                if (cnt <= a.max_verts) K = cnt;
            }
            if (K < 0) {
                if (lane == 0) { a.status[job] = 4; a.num_verts[job] = 0; a.mass[job] = 0; }
                continue;
            }
        } else {
            for (int j = lane; j < M; j += SC_WAVE) L.mark[j] = 1;
            __syncthreads();
        }
        // the kept vertices, ascending: ids + coordinates out (what contact sees is what has mass: the simulated solid is THEIR hull)
        {
            int w = 0;
            for (int base = 0; base < M; base += SC_WAVE) {
                int j = base + lane;
                bool kp = j < M && L.mark[j];
                unsigned long long m = __ballot(kp);
                if (kp) {
                    int d = w + __popcll(m & ((1ull << lane) - 1));
                    int pi = L.rem[j];
                    outi[d] = pi;
                    outv[3 * d] = P[3 * pi]; outv[3 * d + 1] = P[3 * pi + 1]; outv[3 * d + 2] = P[3 * pi + 2];
                }
                w += __popcll(m);
            }
        }
        __threadfence_block();
        __syncthreads();
        F = hull(outv, K, a.eps_rel, maxf, L);
        if (F < 0) {
            if (lane == 0) { a.status[job] = -F; a.num_verts[job] = K; a.mass[job] = 0; }
            continue;
        }
        // mass properties (body_shapes.hull_mass_properties_faces): centre = mean of the hull's vertices, signed tetrahedra about it
        for (int i = lane; i < K; i += SC_WAVE) L.mark[i] = 0;
        __syncthreads();
        for (int f = lane; f < F; f += SC_WAVE) { Face fc = L.face[f]; L.mark[fc.a] = 1; L.mark[fc.b] = 1; L.mark[fc.c] = 1; }
        __syncthreads();
        double cx = 0, cy = 0, cz = 0; int cn = 0;
        for (int i = lane; i < K; i += SC_WAVE)
            if (L.mark[i]) { cx += outv[3 * i]; cy += outv[3 * i + 1]; cz += outv[3 * i + 2]; ++cn; }
        cn = wave_sum_i(cn);
        cx = wave_sum(cx) / cn; cy = wave_sum(cy) / cn; cz = wave_sum(cz) / cn;
        const D3 ctr{cx, cy, cz};
        double vol = 0, f1[3] = {0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
        for (int f = lane; f < F; f += SC_WAVE) {
            Face fc = L.face[f];
            D3 pa = sub3(ldp(outv, fc.a), ctr), pb = sub3(ldp(outv, fc.b), ctr), pc = sub3(ldp(outv, fc.c), ctr);
            double det = dot3(pa, cross3(pb, pc));
            D3 s{pa.x + pb.x + pc.x, pa.y + pb.y + pc.y, pa.z + pb.z + pc.z};
            vol += det;
            f1[0] += det * s.x; f1[1] += det * s.y; f1[2] += det * s.z;
            s2[0] += det * (pa.x * pa.x + pb.x * pb.x + pc.x * pc.x + s.x * s.x);
            s2[1] += det * (pa.x * pa.y + pb.x * pb.y + pc.x * pc.y + s.x * s.y);
            s2[2] += det * (pa.x * pa.z + pb.x * pb.z + pc.x * pc.z + s.x * s.z);
            s2[3] += det * (pa.y * pa.y + pb.y * pb.y + pc.y * pc.y + s.y * s.y);
            s2[4] += det * (pa.y * pa.z + pb.y * pb.z + pc.y * pc.z + s.y * s.z);
            s2[5] += det * (pa.z * pa.z + pb.z * pb.z + pc.z * pc.z + s.z * s.z);
        }
        vol = wave_sum(vol) / 6.0;
        for (int k = 0; k < 3; ++k) f1[k] = wave_sum(f1[k]) / 24.0;
        for (int k = 0; k < 6; ++k) s2[k] = wave_sum(s2[k]) / 120.0;
        if (lane == 0) {
            const double m = a.density * vol;
            const double r[3] = {f1[0] / vol, f1[1] / vol, f1[2] / vol};
            const double cxx = a.density * s2[0] - m * r[0] * r[0], cxy = a.density * s2[1] - m * r[0] * r[1], cxz = a.density * s2[2] - m * r[0] * r[2];
            const double cyy = a.density * s2[3] - m * r[1] * r[1], cyz = a.density * s2[4] - m * r[1] * r[2], czz = a.density * s2[5] - m * r[2] * r[2];
            const double tr = cxx + cyy + czz;
            a.mass[job] = m;
            a.com[3 * job] = cx + r[0]; a.com[3 * job + 1] = cy + r[1]; a.com[3 * job + 2] = cz + r[2];
            double* I = a.inertia + 9 * (size_t)job;
            I[0] = tr - cxx; I[1] = -cxy; I[2] = -cxz; I[3] = -cxy; I[4] = tr - cyy; I[5] = -cyz; I[6] = -cxz; I[7] = -cyz; I[8] = tr - czz;
            a.num_verts[job] = K;
            a.status[job] = 0;
        }
    }
}

}  // namespace

size_t shape_compile_lds_bytes(int max_pts, int max_verts) {
    const size_t maxn = (size_t)(max_pts > max_verts ? max_pts : max_verts), maxf = 2 * maxn;
    return maxf * (sizeof(double4) + sizeof(Face) + sizeof(unsigned) + sizeof(unsigned short)) + sizeof(unsigned short) * ((maxn + 3) & ~(size_t)3) + ((maxn + 15) & ~(size_t)15);
}

int launch_shape_compile(int32_t jobs, const double* pts, const int32_t* job_off, int32_t max_pts, const double* dirs, const int32_t* dir_off, int32_t num_tables,
                         double density, int32_t max_verts, double eps_rel, double* mass, double* com, double* inertia, int32_t* num_verts, int32_t* vert_ids,
                         double* verts, int32_t* status, hipStream_t s) {
    if (jobs == 0) return V2P_OK;
    const size_t lds = shape_compile_lds_bytes(max_pts, max_verts);
    if (lds > 160 * 1024 - 1024 || max_pts > 65535) {
        set_error("v2p_shapes_compile: clouds of up to %d points need %zu bytes of LDS per wave (the CU has 160 KB): thin the clouds", max_pts, lds);
        return V2P_ERR_UNSUPPORTED;
    }
    int rc = check_hip(hipFuncSetAttribute((const void*)shape_compile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(shape_compile_kernel)");
    if (rc != V2P_OK) return rc;
    ShapeCompileArgs a{pts, job_off, dirs, dir_off, num_tables, jobs, max_pts, max_verts, density, eps_rel, mass, com, inertia, num_verts, vert_ids, verts, status};
    // persistent waves: as many workgroups as the LDS lets the chip hold (CUs x floor(160 KB / lds)), each looping over jobs
    int per_cu = (int)((160 * 1024) / (lds + 512));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 16) per_cu = 16;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    int grid = cus * per_cu;
    if (grid > jobs) grid = jobs;
    hipLaunchKernelGGL(shape_compile_kernel, dim3(grid), dim3(SC_WAVE), lds, s, a);
    return check_hip(hipGetLastError(), "shape_compile_kernel");
}

}  // namespace v2p
