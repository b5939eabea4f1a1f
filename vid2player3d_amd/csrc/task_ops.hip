// Task-side kernels of the imitation env: everything HumanoidSMPLIM does around the physics
// step (embodied_pose/env/tasks/humanoid_smpl_im.py).  All of them are HBM/L2-bound
// elementwise + gather work; the mapping is one thread per (env, body) so that the
// reference's row-major [N,24,*] tensors are read/written by adjacent lanes, with an LDS
// reduction over the 24 bodies of an env where the reference takes means.
#include <math.h>

#include "v2p_dev.hpp"
#include "phys_common.hpp"

namespace v2p {

constexpr int EB_BLOCK = 192;          // 8 envs x 24 bodies = 3 wave64
constexpr int ENVS_PER_BLOCK = EB_BLOCK / NB;

#include "post_ops.inc"

// ------------------------------------------------------------------------------------------
// reward  (compute_humanoid_reward, humanoid_smpl_im.py:918-953; dof_to_obs, humanoid_smpl.py:604-635)
// ------------------------------------------------------------------------------------------
struct RewardSpecs {
    float v[8];
};
struct Heights {
    float v[NB];
};

__global__ __launch_bounds__(EB_BLOCK) void reward_kernel(int64_t n, const float* __restrict__ body_pos, const float* __restrict__ body_rot,
                                                          const float* __restrict__ tgt_pos, const float* __restrict__ tgt_rot,
                                                          const float* __restrict__ dof_pos, const float* __restrict__ dof_vel,
                                                          const float* __restrict__ tgt_dof_pos, const float* __restrict__ tgt_dof_vel,
                                                          Heights w, RewardSpecs specs, float* __restrict__ rew, float* __restrict__ sub) {
    __shared__ float red[ENVS_PER_BLOCK][NB][4];
    int le = threadIdx.x / NB, j = threadIdx.x % NB;
    int64_t e = (int64_t)blockIdx.x * ENVS_PER_BLOCK + le;
    RewardPartial r{0.f, 0.f, 0.f, 0.f};
    if (e < n) {
        int64_t bj = e * NB + j;
        V3 z{0.f, 0.f, 0.f};
        int64_t dj = e * NDOF + 3 * (j - 1);
        r = reward_partial(j, ld3(body_pos + bj * 3), ld4(body_rot + bj * 4), ld3(tgt_pos + bj * 3), ld4(tgt_rot + bj * 4),
                           j ? ld3(dof_pos + dj) : z, j ? ld3(dof_vel + dj) : z, j ? ld3(tgt_dof_pos + dj) : z, j ? ld3(tgt_dof_vel + dj) : z,
                           w.v[j]);
    }
    red[le][j][0] = r.dof; red[le][j][1] = r.vel; red[le][j][2] = r.pos; red[le][j][3] = r.rot;
    __syncthreads();
    if (j == 0 && e < n) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < NB; ++b)
            for (int k = 0; k < 4; ++k) s[k] += red[le][b][k];
        float rw, sb[4];
        reward_finish(specs.v, s[0], s[1], s[2], s[3], rw, sb);
        rew[e] = rw;
        for (int k = 0; k < 4; ++k) sub[e * 4 + k] = sb[k];
    }
}

int launch_reward(int64_t n, const float* body_pos, const float* body_rot, const float* tgt_pos, const float* tgt_rot, const float* dof_pos,
                  const float* dof_vel, const float* tgt_dof_pos, const float* tgt_dof_vel, const float* w, const float* specs, float* rew,
                  float* sub, hipStream_t s) {
    if (n <= 0) return V2P_OK;
    Heights hw;
    RewardSpecs sp;
    for (int i = 0; i < NB; ++i) hw.v[i] = w[i];
    for (int i = 0; i < 8; ++i) sp.v[i] = specs[i];
    unsigned blocks = (unsigned)((n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK);
    hipLaunchKernelGGL(reward_kernel, dim3(blocks), dim3(EB_BLOCK), 0, s, n, body_pos, body_rot, tgt_pos, tgt_rot, dof_pos, dof_vel,
                       tgt_dof_pos, tgt_dof_vel, hw, sp, rew, sub);
    return check_hip(hipGetLastError(), "reward_kernel");
}

// ------------------------------------------------------------------------------------------
// reset flags  (compute_humanoid_reset, humanoid_smpl_im.py:956-987)
// ------------------------------------------------------------------------------------------
__global__ void reset_flags_kernel(int64_t n, const int64_t* __restrict__ progress, const float* __restrict__ rb_pos, Heights h,
                                   const float* __restrict__ cur_time, const float* __restrict__ clip_len, float max_len, int early,
                                   int64_t* __restrict__ reset_out, int64_t* __restrict__ term_out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    int64_t prog = progress[e];
    int64_t term = 0;
    if (early) {
        bool fall = false;
        for (int b = 0; b < NB; ++b) fall |= rb_pos[(e * NB + b) * 3 + 2] < h.v[b];
        if (fall && prog > 1) term = 1;
    }
    bool cond = ((float)prog >= max_len - 1.f) || (cur_time[e] >= clip_len[e]);
    reset_out[e] = cond ? 1 : term;
    term_out[e] = term;
}

int launch_reset_flags(int64_t n, const int64_t* progress, const float* rb_pos, const float* heights, const float* cur_time,
                       const float* clip_len, float max_len, int early, int64_t* reset_out, int64_t* term_out, hipStream_t s) {
    if (n <= 0) return V2P_OK;
    Heights h;
    for (int i = 0; i < NB; ++i) h.v[i] = heights[i];
    unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(reset_flags_kernel, dim3(blocks), dim3(256), 0, s, n, progress, rb_pos, h, cur_time, clip_len, max_len, early,
                       reset_out, term_out);
    return check_hip(hipGetLastError(), "reset_flags_kernel");
}

// ------------------------------------------------------------------------------------------
// 734-d in-network imitation observation  (humanoid_smpl_im.py:773-850, next row f-1)
// layout: root_h 1 | local_body_pos 69 | local_body_rot 144 | local_vel 72 | local_ang_vel 72 | dof_vel 69 |
//         rel_root_h 1 | rel_root_rot 6 | rel_2d_pos 2 | rel_heading 2 | rel_dof 69 | rel_body_pos 72 | rel_body_rot 144 | motion_bodies 11
// ------------------------------------------------------------------------------------------
// Row r of the inputs: element pointers + row strides (floats), so that the same kernel reads either nine separate contiguous
// tensors or the packed 461-d observation row + the 378-d context frame the network is handed (im_network_builder.py:150-189).
struct ObsImSrc {
    const float *body_pos, *body_rot, *dof_pos, *dof_vel, *body_vel, *body_ang_vel, *motion_bodies;  // per row
    int64_t s_body_pos, s_body_rot, s_dof_pos, s_dof_vel, s_body_vel, s_body_ang_vel, s_motion_bodies;
    const float *tgt_pos, *tgt_rot, *tgt_dof_pos;  // per row, or (steps > 1 / context) per (env, frame)
    int64_t s_tgt_pos, s_tgt_rot, s_tgt_dof_pos;   // stride between FRAMES of the target
    int64_t steps, ctx_frames, first_frame;         // row r = env * steps + k reads target frame env * ctx_frames + first_frame + k
};

__global__ __launch_bounds__(EB_BLOCK) void obs_imitation_kernel(int64_t n, ObsImSrc in, const float* __restrict__ nmean,
                                                                 const float* __restrict__ nstd, float nclip, float* __restrict__ obs) {
    int le = threadIdx.x / NB, j = threadIdx.x % NB;
    int64_t e = (int64_t)blockIdx.x * ENVS_PER_BLOCK + le;
    if (e >= n) return;
    const int64_t tf = (e / in.steps) * in.ctx_frames + in.first_frame + e % in.steps;  // target frame of this row
    const float* body_pos = in.body_pos + e * in.s_body_pos;
    const float* body_rot = in.body_rot + e * in.s_body_rot;
    const float* tgt_pos = in.tgt_pos + tf * in.s_tgt_pos;
    const float* tgt_rot = in.tgt_rot + tf * in.s_tgt_rot;
    float* o = obs + e * 734;
    V3 root_pos = ld3(body_pos);
    Q4 root_rot = ref_remove_base_rot(ld4(body_rot));
    float heading = ref_calc_heading(root_rot);
    Q4 hinv = ref_heading_quat(-heading);
    V3 p = ld3(body_pos + j * 3);
    Q4 q = ld4(body_rot + j * 4);
    V3 lp = ref_quat_rotate(hinv, p - root_pos);
    if (j > 0) st3(o + 1 + 3 * (j - 1), lp);
    V3 tn, nm;
    if (j == 0) ref_quat_to_tan_norm(root_rot, tn, nm);  // reference overwrites the root entry with the un-headed root (:807-808)
    else ref_quat_to_tan_norm(qmul(hinv, q), tn, nm);
    st3(o + 70 + 6 * j, tn);
    st3(o + 70 + 6 * j + 3, nm);
    st3(o + 214 + 3 * j, ref_quat_rotate(hinv, ld3(in.body_vel + e * in.s_body_vel + j * 3)));
    st3(o + 286 + 3 * j, ref_quat_rotate(hinv, ld3(in.body_ang_vel + e * in.s_body_ang_vel + j * 3)));
    V3 tp = ld3(tgt_pos + j * 3);
    Q4 tq = ld4(tgt_rot + j * 4);
    st3(o + 507 + 3 * j, ref_quat_rotate(hinv, tp - p));
    ref_quat_to_tan_norm(qmul(qconj(q), tq), tn, nm);
    st3(o + 579 + 6 * j, tn);
    st3(o + 579 + 6 * j + 3, nm);
    if (j > 0) {
        const int dj = 3 * (j - 1);
        st3(o + 358 + dj, ld3(in.dof_vel + e * in.s_dof_vel + dj));
        st3(o + 438 + dj, ld3(in.tgt_dof_pos + tf * in.s_tgt_dof_pos + dj) - ld3(in.dof_pos + e * in.s_dof_pos + dj));
    } else {
        o[0] = root_pos.z;
        V3 t_root_pos = tp;
        Q4 t_root_rot = ref_remove_base_rot(tq);
        o[427] = root_pos.z - t_root_pos.z;
        ref_quat_to_tan_norm(qmul(t_root_rot, qconj(root_rot)), tn, nm);
        st3(o + 428, tn);
        st3(o + 431, nm);
        V3 rel = ref_quat_rotate(hinv, t_root_pos - root_pos);
        o[434] = rel.x; o[435] = rel.y;
        float dh = ref_calc_heading(t_root_rot) - heading;
        o[436] = cosf(dh); o[437] = sinf(dh);
    }
    if (j < 11) o[723 + j] = in.motion_bodies[e * in.s_motion_bodies + j];
    if (nmean) {
        // RunningNorm (eval) fused: the row was just written by this workgroup's 24 threads of env e
        __syncthreads();
        for (int k = j; k < 734; k += NB) {
            float y = (o[k] - nmean[k]) / (nstd[k] + 1e-8f);
            o[k] = fminf(fmaxf(y, -nclip), nclip);
        }
    }
}

static int launch_obs_imitation_src(int64_t n, const ObsImSrc& in, const float* nmean, const float* nstd, float nclip, float* obs, hipStream_t s) {
    if (n <= 0) return V2P_OK;
    unsigned blocks = (unsigned)((n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK);
    hipLaunchKernelGGL(obs_imitation_kernel, dim3(blocks), dim3(EB_BLOCK), 0, s, n, in, nmean, nstd, nclip, obs);
    return check_hip(hipGetLastError(), "obs_imitation_kernel");
}

int launch_obs_imitation(int64_t n, const float* body_pos, const float* body_rot, const float* tgt_pos, const float* tgt_rot,
                         const float* dof_pos, const float* dof_vel, const float* tgt_dof_pos, const float* body_vel,
                         const float* body_ang_vel, const float* motion_bodies, const float* nmean, const float* nstd, float nclip, float* obs,
                         hipStream_t s) {
    ObsImSrc in = {body_pos, body_rot, dof_pos, dof_vel, body_vel, body_ang_vel, motion_bodies, NB * 3, NB * 4, NDOF, NDOF, NB * 3, NB * 3, 11,
                   tgt_pos, tgt_rot, tgt_dof_pos, NB * 3, NB * 4, NDOF, 1, 1, 0};
    return launch_obs_imitation_src(n, in, nmean, nstd, nclip, obs, s);
}

// the network's own inputs: obs rows [rows,461] in the order of humanoid_smpl_im.py:198 and context frames [envs,ctx_frames,378] in the
// order of :202 (body_pos 72 | body_rot 96 | dof_pos 69 | ...); rows = envs * steps, row env*steps+k pairs with frame first_frame+k
int launch_obs_imitation_packed(int64_t rows, int64_t steps, const float* obs461, const float* context_feat, int64_t ctx_frames, int64_t first_frame,
                                const float* nmean, const float* nstd, float nclip, float* obs, hipStream_t s) {
    ObsImSrc in = {obs461, obs461 + 72, obs461 + 168, obs461 + 237, obs461 + 306, obs461 + 378, obs461 + 450, NOBS, NOBS, NOBS, NOBS, NOBS, NOBS, NOBS,
                   context_feat, context_feat + 72, context_feat + 168, V2P_CONTEXT_DIM, V2P_CONTEXT_DIM, V2P_CONTEXT_DIM, steps, ctx_frames, first_frame};
    return launch_obs_imitation_src(rows, in, nmean, nstd, nclip, obs, s);
}

// ------------------------------------------------------------------------------------------
// GAE reverse scan (CommonAgent.discount_values, learning/common_agent.py:423-435): one thread per env, T steps
// ------------------------------------------------------------------------------------------
__global__ void gae_kernel(int64_t horizon, int64_t n, const float* __restrict__ fdones, const float* __restrict__ values,
                           const float* __restrict__ rewards, const float* __restrict__ next_values, float gamma, float tau,
                           float* __restrict__ advs) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float last = 0.f;
    for (int64_t t = horizon - 1; t >= 0; --t) {
        const int64_t i = t * n + e;
        float not_done = 1.f - fdones[i];
        float delta = rewards[i] + gamma * next_values[i] - values[i];
        last = delta + gamma * tau * not_done * last;
        advs[i] = last;
    }
}

int launch_gae(int64_t horizon, int64_t n, const float* fdones, const float* values, const float* rewards, const float* next_values, float gamma,
               float tau, float* advs, hipStream_t s) {
    if (n <= 0 || horizon <= 0) return V2P_OK;
    unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(gae_kernel, dim3(blocks), dim3(256), 0, s, horizon, n, fdones, values, rewards, next_values, gamma, tau, advs);
    return check_hip(hipGetLastError(), "gae_kernel");
}

// ------------------------------------------------------------------------------------------
// Bookkeeping of one rollout step AFTER env.step (ImitatorAgent.play_steps, agents/im_agent.py:380-409): what the reference does with
// ~25 small torch ops and two .nonzero() host syncs per step - rewards / dones / next_obses into the experience buffer, the dones and
// terminate flags as floats, the running episode returns and lengths, and the episode statistics (here: device accumulators in float64)
// - in ONE launch: workgroups [0, env_blocks) take 256 envs each (scalars + a block reduction -> 9 atomic adds), the rest stream the
// observation rows into the buffer.
//   dones = reset.float(); terminated = terminate.float(); cur_rewards += rew; cur_lengths += 1
//   step_dones = dones (1 - prev_dones); alive_before = 1 - prev_dones
//   acc[0] += sum step_dones; acc[1] += sum cur_rewards step_dones; acc[2] += sum cur_lengths step_dones; acc[3] += sum alive_before;
//   acc[4] += sum rew alive_before; sub_acc[k] += sum sub_rewards[:, k] alive_before; prev_dones = dones
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rollout_record_kernel(int64_t n, int env_blocks, const float* __restrict__ obs, int64_t obs_dim, const float* __restrict__ rew,
                                                             const int64_t* __restrict__ reset, const int64_t* __restrict__ terminate,
                                                             const float* __restrict__ sub_rewards, float* __restrict__ next_obs_row,
                                                             float* __restrict__ rewards_row, float* __restrict__ dones_row, float* __restrict__ dones,
                                                             float* __restrict__ terminated, float* __restrict__ prev_dones, float* __restrict__ cur_rewards,
                                                             float* __restrict__ cur_lengths, double* __restrict__ acc, double* __restrict__ sub_acc) {
    if ((int)blockIdx.x >= env_blocks) {
        // the observation rows: a flat, coalesced copy (16-byte pieces when both sides allow it)
        const int64_t total = n * obs_dim, nb = gridDim.x - env_blocks, bi = blockIdx.x - env_blocks;
        if ((((uintptr_t)obs | (uintptr_t)next_obs_row) & 15) == 0 && (total & 3) == 0) {
            const float4* s4 = (const float4*)obs;
            float4* d4 = (float4*)next_obs_row;
            for (int64_t i = bi * 256 + threadIdx.x; i < total / 4; i += nb * 256) d4[i] = s4[i];
        } else {
            for (int64_t i = bi * 256 + threadIdx.x; i < total; i += nb * 256) next_obs_row[i] = obs[i];
        }
        return;
    }
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (e < n) {
        const float r = rew[e], d = (float)reset[e], pd = prev_dones[e];
        const float cr = cur_rewards[e] + r, cl = cur_lengths[e] + 1.f;
        const float sd = d * (1.f - pd), alive = 1.f - pd;
        rewards_row[e] = r;
        dones_row[e] = d;
        dones[e] = d;
        terminated[e] = (float)terminate[e];
        cur_rewards[e] = cr;
        cur_lengths[e] = cl;
        prev_dones[e] = d;
        v[0] = sd; v[1] = cr * sd; v[2] = cl * sd; v[3] = alive; v[4] = r * alive;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[5 + k] = sub_rewards[e * 4 + k] * alive;
    }
    __shared__ double red[4][9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 9; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 9) {
        const double s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (threadIdx.x < 5) atomicAdd(acc + threadIdx.x, s);
        else atomicAdd(sub_acc + (threadIdx.x - 5), s);
    }
}

// The critic's output of a rollout step into the experience buffer (im_agent.py:292-303, 355, 398): un-normalised with the value normaliser
// (rl_games RunningMeanStd, unnorm: sqrt(var + eps) * clamp(x, -5, 5) + mean, float32 like torch evaluates it), written as `values` of step n
// and - masked by the terminations of the step that produced the observation (end_value_type 'next') - as `next_values` of step n - 1.
__global__ void value_record_kernel(int64_t n, const float* __restrict__ x, const double* __restrict__ mean, const double* __restrict__ var, float eps,
                                    const float* __restrict__ terminated, float* __restrict__ values_row, float* __restrict__ next_values_row) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float y = x[e];
    if (mean) {
        const float scale = sqrtf((float)var[0] + eps);
        y = scale * fminf(fmaxf(y, -5.f), 5.f) + (float)mean[0];
    }
    if (values_row) values_row[e] = y;
    if (next_values_row) next_values_row[e] = y * (1.f - terminated[e]);
}

int launch_value_record(int64_t n, const float* x, const double* mean, const double* var, float eps, const float* terminated, float* values_row,
                        float* next_values_row, hipStream_t s) {
    if (n <= 0) return V2P_OK;
    hipLaunchKernelGGL(value_record_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, x, mean, var, eps, terminated, values_row, next_values_row);
    return check_hip(hipGetLastError(), "value_record_kernel");
}

int launch_rollout_record(int64_t n, const float* obs, int64_t obs_dim, const float* rew, const int64_t* reset, const int64_t* terminate, const float* sub_rewards,
                          float* next_obs_row, float* rewards_row, float* dones_row, float* dones, float* terminated, float* prev_dones, float* cur_rewards,
                          float* cur_lengths, double* acc, double* sub_acc, hipStream_t s) {
    if (n <= 0) return V2P_OK;
    const int env_blocks = (int)((n + 255) / 256);
    const int64_t total = n * obs_dim;
    int copy_blocks = next_obs_row ? (int)((total / 4 + 1023) / 1024) : 0;  // ~4 float4 per thread
    if (copy_blocks > 4096) copy_blocks = 4096;
    hipLaunchKernelGGL(rollout_record_kernel, dim3(env_blocks + copy_blocks), dim3(256), 0, s, n, env_blocks, obs, obs_dim, rew, reset, terminate, sub_rewards,
                       next_obs_row, rewards_row, dones_row, dones, terminated, prev_dones, cur_rewards, cur_lengths, acc, sub_acc);
    return check_hip(hipGetLastError(), "rollout_record_kernel");
}

// ------------------------------------------------------------------------------------------
// policy head of the rollout (models/im_network_builder.py:219-228 `eval_actor`: residual action, mu[:, :69] += the context's dof_pos of
// step t; models/im_models.py:45-48: action = Normal(mu, sigma).sample(), neglogp): one wave per env, the 75 action components on lanes
// k and k + 64, the two sums of neglogp as wave reductions.  `noise` is the standard-normal draw (torch's generator stays the source
// of randomness), `mu` holds the network's output on entry and the residual mean on exit.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void policy_head_kernel(int64_t n, float* __restrict__ mu, const float* __restrict__ context_feat, int64_t ctx_frames,
                                                          int64_t frame, const float* __restrict__ logstd, const float* __restrict__ noise,
                                                          float* __restrict__ action, float* __restrict__ sigma_out, float* __restrict__ neglogp,
                                                          float* __restrict__ action_row, float* __restrict__ mu_row) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= n) return;
    const float* tgt = context_feat + (e * ctx_frames + frame) * V2P_CONTEXT_DIM + 168;  // dof_pos block of the context frame
    float quad = 0.f, lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int k = lane + 64 * r;
        if (k < NACT) {
            const int64_t i = e * NACT + k;
            float m = mu[i];
            if (k < NDOF) m += tgt[k];
            const float ls = logstd[k], sg = expf(ls);
            const float a = m + sg * noise[i];
            const float z = (a - m) / sg;
            mu[i] = m;
            action[i] = a;
            if (sigma_out) sigma_out[i] = sg;
            // (the experience buffer's rows, written here instead of by copy kernels behind this one; `action` stays a tensor of its own:
            // env.step masks it in place)
            if (action_row) action_row[i] = a;
            if (mu_row) mu_row[i] = m;
            quad += z * z;
            lsum += ls;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { quad += __shfl_xor(quad, o); lsum += __shfl_xor(lsum, o); }
    if (lane == 0) neglogp[e] = 0.5f * quad + 0.5f * 1.8378770664093453f * (float)NACT + lsum;  // log(2 pi)
}

int launch_policy_head(int64_t n, float* mu, const float* context_feat, int64_t ctx_frames, int64_t frame, const float* logstd, const float* noise,
                       float* action, float* sigma_out, float* neglogp, hipStream_t s, float* action_row, float* mu_row) {
    if (n <= 0) return V2P_OK;
    hipLaunchKernelGGL(policy_head_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, n, mu, context_feat, ctx_frames, frame, logstd, noise, action,
                       sigma_out, neglogp, action_row, mu_row);
    return check_hip(hipGetLastError(), "policy_head_kernel");
}

// ------------------------------------------------------------------------------------------
// env kernels
// ------------------------------------------------------------------------------------------
struct EnvView {
    v2p_env_buffers b;
    v2p_motion_tables t;
    EnvParams p;
    const int64_t* motion_id;
    float* state;
    float* ctrl;
    float* out;
    int64_t n;
    int cur;  // current target buffer
    PairView pair;  // pre-physics only: scatter of the pairing order for the next physics launch (perm NULL = none)
};

static EnvView make_view(const v2p_env* e) {
    EnvView v;
    v.b = e->buf;
    v.t = e->mlib->t;
    v.p = e->p;
    v.motion_id = e->motion_id;
    v.state = e->state;
    v.ctrl = e->ctrl;
    v.out = e->out;
    v.n = e->n;
    v.cur = e->cur_target;
    v.pair = PairView{nullptr, nullptr, nullptr, nullptr, 0, 0};
    return v;
}

// raw-state observation row of env e, body j
__device__ __forceinline__ void write_obs(const EnvView& v, int64_t e, int j, V3 pos, Q4 rot, V3 vel, V3 ang, V3 dpos, V3 dvel) {
    write_obs_row(v.b.obs + e * NOBS, v.t.motion_bodies + v.motion_id[e] * 11, j, pos, rot, vel, ang, dpos, dvel);
}

// ---- reset: reference-state init (humanoid_smpl_im.py:489-528, 741-755; humanoid_smpl.py:153-173)
__global__ __launch_bounds__(EB_BLOCK) void env_reset_kernel(EnvView v, const int64_t* __restrict__ env_ids, int64_t n,
                                                             const float* __restrict__ motion_times) {
    int le = threadIdx.x / NB, j = threadIdx.x % NB;
    int64_t i = (int64_t)blockIdx.x * ENVS_PER_BLOCK + le;
    if (i >= n) return;
    int64_t e = env_ids ? env_ids[i] : i;
    int64_t mid = v.motion_id[e];
    float t0 = motion_times[i];
    FrameRef fr = frame_lookup(v.t, mid, t0, 1, v.p.ground_tolerance);
    BodySample s = sample_body_values(v.t, fr, j);
    // exposed tensors
    float* rb = v.b.rb_state + (e * NB + j) * 13;
    st3(rb, s.pos);
    st4(rb + 3, s.rot);
    V3 zero{0.f, 0.f, 0.f};
    st3(rb + 7, zero);   // body velocities are zeroed on reset (_set_env_state :748-749)
    st3(rb + 10, zero);
    if (j == 0) {
        float* r = v.b.root_states + e * 13;
        st3(r, s.pos); st4(r + 3, s.rot); st3(r + 7, s.root_vel); st3(r + 10, s.root_ang_vel);
        v.state[SIDX(ST_ROOT_POS + 0)] = s.pos.x; v.state[SIDX(ST_ROOT_POS + 1)] = s.pos.y; v.state[SIDX(ST_ROOT_POS + 2)] = s.pos.z;
        Q4 rq = qnormalize(s.rot);
        v.state[SIDX(ST_ROOT_QUAT + 0)] = rq.x; v.state[SIDX(ST_ROOT_QUAT + 1)] = rq.y;
        v.state[SIDX(ST_ROOT_QUAT + 2)] = rq.z; v.state[SIDX(ST_ROOT_QUAT + 3)] = rq.w;
        v.state[SIDX(ST_VEL + 0)] = s.root_vel.x; v.state[SIDX(ST_VEL + 1)] = s.root_vel.y; v.state[SIDX(ST_VEL + 2)] = s.root_vel.z;
        v.state[SIDX(ST_VEL + 3)] = s.root_ang_vel.x; v.state[SIDX(ST_VEL + 4)] = s.root_ang_vel.y; v.state[SIDX(ST_VEL + 5)] = s.root_ang_vel.z;
        v.b.cur_time[e] = t0;
        v.b.reset_time[e] = t0;
        v.b.progress[e] = 0;
        v.b.reset[e] = 0;
        v.b.terminate[e] = 0;
    } else {
        float* d = v.b.dof_state + (e * NDOF + 3 * (j - 1)) * 2;
        d[0] = s.dof_pos.x; d[1] = s.dof_vel.x; d[2] = s.dof_pos.y; d[3] = s.dof_vel.y; d[4] = s.dof_pos.z; d[5] = s.dof_vel.z;
        // gym.set_dof_state_tensor_indexed: the engine's joint quaternion is rebuilt from the exp-map dof_pos
        Q4 jq = ref_exp_map_to_quat(s.dof_pos);
        int base = ST_JQUAT + 4 * (j - 1);
        v.state[SIDX(base + 0)] = jq.x; v.state[SIDX(base + 1)] = jq.y; v.state[SIDX(base + 2)] = jq.z; v.state[SIDX(base + 3)] = jq.w;
        int vb = ST_VEL + 6 + 3 * (j - 1);
        v.state[SIDX(vb + 0)] = s.dof_vel.x; v.state[SIDX(vb + 1)] = s.dof_vel.y; v.state[SIDX(vb + 2)] = s.dof_vel.z;
    }
    write_obs(v, e, j, s.pos, s.rot, zero, zero, s.dof_pos, s.dof_vel);
    // target = state one control step ahead (_set_target_motion_state :594-624)
    FrameRef fr1 = frame_lookup(v.t, mid, t0 + v.p.dt, 1, v.p.ground_tolerance);
    sample_body(v.t, fr1, j, e, packed_out(v.b.target[v.cur]));
}

// ---- context window (_init_context, humanoid_smpl_im.py:530-563)
__global__ __launch_bounds__(EB_BLOCK) void env_context_kernel(EnvView v, const int64_t* __restrict__ env_ids, int64_t n,
                                                               const float* __restrict__ motion_times) {
    const int W = v.p.context_length + 2 * v.p.context_padding;
    int lq = threadIdx.x / NB, j = threadIdx.x % NB;
    int64_t q = (int64_t)blockIdx.x * ENVS_PER_BLOCK + lq;
    if (q >= n * W) return;
    int64_t i = q / W;
    int w = (int)(q - i * W);
    int64_t e = env_ids ? env_ids[i] : i;
    int64_t mid = v.motion_id[e];
    float t = (motion_times[i] + v.p.dt) + v.p.dt * (float)(w - v.p.context_padding);
    FrameRef fr = frame_lookup(v.t, mid, t, 1, v.p.ground_tolerance);
    BodySample s = sample_body_values(v.t, fr, j);
    float* o = v.b.context_feat + (e * W + w) * V2P_CONTEXT_DIM;
    st3(o + 3 * j, s.pos);
    st4(o + 72 + 4 * j, s.rot);
    st3(o + 237 + 3 * j, s.pos);
    if (j > 0) {
        st3(o + 168 + 3 * (j - 1), s.dof_pos);
        st3(o + 309 + 3 * (j - 1), s.dof_pos);
    } else if (v.b.context_mask) {
        v.b.context_mask[e * W + w] = t <= v.t.motion_lengths[mid] + 2.f * v.p.dt ? 1 : 0;
    }
}

int launch_env_reset(v2p_env* env, const int64_t* env_ids, int64_t n, const float* motion_times, hipStream_t s) {
    if (n <= 0) return V2P_OK;
    EnvView v = make_view(env);
    unsigned blocks = (unsigned)((n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK);
    hipLaunchKernelGGL(env_reset_kernel, dim3(blocks), dim3(EB_BLOCK), 0, s, v, env_ids, n, motion_times);
    int rc = check_hip(hipGetLastError(), "env_reset_kernel");
    if (rc) return rc;
    if (env->buf.context_feat) rc = launch_env_context(env, env_ids, n, motion_times, s);
    return rc;
}

int launch_env_context(v2p_env* env, const int64_t* env_ids, int64_t n, const float* motion_times, hipStream_t s) {
    if (n <= 0) return V2P_OK;
    EnvView v = make_view(env);
    int64_t q = n * (env->p.context_length + 2 * env->p.context_padding);
    unsigned cb = (unsigned)((q + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK);
    hipLaunchKernelGGL(env_context_kernel, dim3(cb), dim3(EB_BLOCK), 0, s, v, env_ids, n, motion_times);
    return check_hip(hipGetLastError(), "env_context_kernel");
}

// (pre-physics, humanoid_smpl_im.py:125-157, lives in physics_ll.hip: one implementation, compiled once with precise semantics,
// serves both the stand-alone env_pre_kernel and the prologue of the physics kernel)

// ---- export: physics outputs (structure-of-arrays) -> the row-major tensors the reference exposes
//      (the six gym.refresh_*_tensor calls, humanoid_smpl_im.py:452-468)
__global__ __launch_bounds__(EB_BLOCK) void env_export_kernel(EnvView v) {
    int le = threadIdx.x / NB, j = threadIdx.x % NB;
    int64_t e = (int64_t)blockIdx.x * ENVS_PER_BLOCK + le;
    if (e >= v.n) return;
    float* rb = v.b.rb_state + (e * NB + j) * 13;
#pragma unroll
    for (int k = 0; k < 13; ++k) rb[k] = v.out[OIDX(OUT_RB + j * 13 + k)];
    float* cf = v.b.contact_force + (e * NB + j) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) cf[k] = v.out[OIDX(OUT_CONTACT + j * 3 + k)];
    if (j == 0) {
        float* r = v.b.root_states + e * 13;
#pragma unroll
        for (int k = 0; k < 13; ++k) r[k] = v.out[OIDX(OUT_RB + k)];
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int d = 3 * (j - 1) + k;
            v.b.dof_state[(e * NDOF + d) * 2] = v.out[OIDX(OUT_DOF_POS + d)];
            v.b.dof_state[(e * NDOF + d) * 2 + 1] = v.state[SIDX(ST_VEL + 6 + d)];
            v.b.dof_force[e * NDOF + d] = v.out[OIDX(OUT_DOF_FORCE + d)];
        }
    }
}

int launch_env_export(v2p_env* env, hipStream_t s) {
    EnvView v = make_view(env);
    unsigned blocks = (unsigned)((env->n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK);
    hipLaunchKernelGGL(env_export_kernel, dim3(blocks), dim3(EB_BLOCK), 0, s, v);
    return check_hip(hipGetLastError(), "env_export_kernel");
}

// ---- push caller-edited root/dof state into the engine (set_*_state_tensor_indexed, humanoid_smpl.py:161-173)
__global__ __launch_bounds__(EB_BLOCK) void env_push_state_kernel(EnvView v, const int64_t* __restrict__ env_ids, int64_t n) {
    int le = threadIdx.x / NB, j = threadIdx.x % NB;
    int64_t i = (int64_t)blockIdx.x * ENVS_PER_BLOCK + le;
    if (i >= n) return;
    int64_t e = env_ids ? env_ids[i] : i;
    if (j == 0) {
        const float* r = v.b.root_states + e * 13;
        for (int k = 0; k < 3; ++k) v.state[SIDX(ST_ROOT_POS + k)] = r[k];
        Q4 q = qnormalize(ld4(r + 3));
        v.state[SIDX(ST_ROOT_QUAT + 0)] = q.x; v.state[SIDX(ST_ROOT_QUAT + 1)] = q.y;
        v.state[SIDX(ST_ROOT_QUAT + 2)] = q.z; v.state[SIDX(ST_ROOT_QUAT + 3)] = q.w;
        for (int k = 0; k < 6; ++k) v.state[SIDX(ST_VEL + k)] = r[7 + k];
    } else {
        const float* d = v.b.dof_state + (e * NDOF + 3 * (j - 1)) * 2;
        Q4 jq = ref_exp_map_to_quat(V3{d[0], d[2], d[4]});
        int base = ST_JQUAT + 4 * (j - 1);
        v.state[SIDX(base + 0)] = jq.x; v.state[SIDX(base + 1)] = jq.y; v.state[SIDX(base + 2)] = jq.z; v.state[SIDX(base + 3)] = jq.w;
        int vb = ST_VEL + 6 + 3 * (j - 1);
        v.state[SIDX(vb + 0)] = d[1]; v.state[SIDX(vb + 1)] = d[3]; v.state[SIDX(vb + 2)] = d[5];
    }
}

int launch_env_push_state(v2p_env* env, const int64_t* env_ids, int64_t n, int /*with_rb*/, hipStream_t s) {
    if (n <= 0) return V2P_OK;
    EnvView v = make_view(env);
    unsigned blocks = (unsigned)((n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK);
    hipLaunchKernelGGL(env_push_state_kernel, dim3(blocks), dim3(EB_BLOCK), 0, s, v, env_ids, n);
    return check_hip(hipGetLastError(), "env_push_state_kernel");
}

// ---- post-physics (humanoid_smpl_im.py:398-418): progress/time, new target, obs, reward, reset
__global__ __launch_bounds__(EB_BLOCK) void env_post_kernel(EnvView v) {
    __shared__ float red[ENVS_PER_BLOCK][NB][4];
    __shared__ int fell[ENVS_PER_BLOCK];
    int le = threadIdx.x / NB, j = threadIdx.x % NB;
    int64_t e = (int64_t)blockIdx.x * ENVS_PER_BLOCK + le;
    bool live = e < v.n;
    if (j == 0) fell[le] = 0;
    __syncthreads();
    RewardPartial r{0.f, 0.f, 0.f, 0.f};
    float t_new = 0.f;
    int64_t mid = 0;
    if (live && j == 0 && v.pair.perm) pair_scatter(v.pair, e);  // wave order of the next physics launch (see physics_ll.hip)
    bool fl = false;
    if (live) {
        mid = v.motion_id[e];
        t_new = v.b.cur_time[e] + v.p.dt;  // _cur_ref_motion_times += dt
        // sim state as exposed
        const float* rb = v.b.rb_state + (e * NB + j) * 13;
        V3 pos = ld3(rb), vel = ld3(rb + 7), ang = ld3(rb + 10);
        Q4 rot = ld4(rb + 3);
        V3 dpos{0.f, 0.f, 0.f}, dvel{0.f, 0.f, 0.f};
        if (j > 0) {
            const float* d = v.b.dof_state + (e * NDOF + 3 * (j - 1)) * 2;
            dpos = V3{d[0], d[2], d[4]};
            dvel = V3{d[1], d[3], d[5]};
        }
        r = post_body(v.b, v.t, v.p, mid, t_new, v.cur, e, j, pos, rot, vel, ang, dpos, dvel, fl);
        if (fl) atomicOr(&fell[le], 1);
    }
    red[le][j][0] = r.dof; red[le][j][1] = r.vel; red[le][j][2] = r.pos; red[le][j][3] = r.rot;
    __syncthreads();
    if (live && j == 0) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < NB; ++b)
            for (int k = 0; k < 4; ++k) s[k] += red[le][b][k];
        post_env(v.b, v.t, v.p, mid, t_new, e, s, fell[le] != 0);
    }
}

int launch_env_post(v2p_env* env, hipStream_t s) {
    EnvView v = make_view(env);
    unsigned blocks = (unsigned)((env->n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK);
    hipLaunchKernelGGL(env_post_kernel, dim3(blocks), dim3(EB_BLOCK), 0, s, v);
    int rc = check_hip(hipGetLastError(), "env_post_kernel");
    if (rc == V2P_OK) env->cur_target = 1 - env->cur_target;
    return rc;
}

}  // namespace v2p
