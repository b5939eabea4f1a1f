// Stand-alone reference-motion sampler kernel: MotionLib.get_motion_state
// (embodied_pose/utils/motion_lib.py:164-266).  HBM-bound gather: per query 2 x 1356 B of
// table rows in, 1324 B out; one thread per (query, body).
#include "v2p_dev.hpp"

namespace v2p {

constexpr int MS_BLOCK = 192;  // 8 queries x 24 bodies = 3 full wave64s

__global__ __launch_bounds__(MS_BLOCK) void motion_state_kernel(v2p_motion_tables t, const int64_t* __restrict__ ids,
                                                                const float* __restrict__ times, int64_t nq, int adjust_height,
                                                                float ground_tol, MSOut o) {
    int64_t tid = (int64_t)blockIdx.x * MS_BLOCK + threadIdx.x;
    int64_t q = tid / NB;
    int j = (int)(tid - q * NB);
    if (q >= nq) return;
    FrameRef fr = frame_lookup(t, ids[q], times[q], adjust_height, ground_tol);
    sample_body(t, fr, j, q, o);
}

int launch_motion_state(const v2p_motion_tables& t, const int64_t* ids, const float* times, int64_t q, int adjust_height, float ground_tol,
                        float* const out[9], hipStream_t s) {
    if (q <= 0) return V2P_OK;
    MSOut o;
    const int64_t nat[9] = {3, 4, NDOF, 3, 3, NDOF, 12, NB * 3, NB * 4};
    for (int i = 0; i < 9; ++i) { o.p[i] = out[i]; o.stride[i] = nat[i]; }
    int64_t threads = q * NB;
    unsigned blocks = (unsigned)((threads + MS_BLOCK - 1) / MS_BLOCK);
    hipLaunchKernelGGL(motion_state_kernel, dim3(blocks), dim3(MS_BLOCK), 0, s, t, ids, times, q, adjust_height, ground_tol, o);
    return check_hip(hipGetLastError(), "motion_state_kernel");
}

}  // namespace v2p
