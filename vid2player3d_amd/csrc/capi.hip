// C-ABI entry points of libv2p_rollout.so (declared in include/v2p_rollout.h).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <new>
#include <vector>

#include "v2p_internal.hpp"
#include "phys_common.hpp"

namespace v2p {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return V2P_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return V2P_ERR_HIP;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace v2p

// The link-per-lane physics kernel is built TWICE from the same source (build.py): the default object (168 VGPRs, three waves per SIMD,
// contact records and phase-dead values parked in LDS) and `physics_ll_regs.o` (-Dv2p=v2p_regs -DV2P_LL_WPS=2 -DV2P_LL_PARK2=0
// -DV2P_LL_PARK3=0: 256 VGPRs, two waves per SIMD, everything in registers).  Where a launch is as long as its heaviest env pair - up to
// ~5000 envs on one GPU - the register build is 7 - 12 % faster (no LDS round trips in the heaviest wave's chain; compiled for ILP), where
// the wave slots are full the LDS build is 15 % faster (profiles/r04e_dual_build.txt).  The second object lives in its own
// namespace; what it calls from this file is forwarded here.  (-Dv2p=v2p_regs renames the namespace in every header that object
// includes as well: the types it sees inside `v2p_env` are `v2p_regs::` twins of this file's - same source, same layout, and only the
// pointer crosses the boundary.)
namespace v2p_regs {
void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    v2p::set_error("%s", buf);
}
int check_hip(hipError_t e, const char* what) { return v2p::check_hip(e, what); }
int launch_env_physics_ll(v2p_env* e, hipStream_t s, float* actions, int* fused_post);
}  // namespace v2p_regs

using namespace v2p;

// Profiling switches (V2P_WAVE_TIMES, V2P_PHASE_TIMING, V2P_PHASE_HEAVY, V2P_ENVS_PER_BLOCK) are read from the environment ONLY in a
// process that sets V2P_DEBUG=1 (tools/*.sh do); every engine option is a field of v2p_sim_cfg.
namespace v2p {
const char* debug_env(const char* name) {
    const char* d = getenv("V2P_DEBUG");
    return (d && d[0] == '1' && d[1] == 0) ? getenv(name) : nullptr;
}
}  // namespace v2p
namespace v2p_regs { const char* debug_env(const char* name) { return v2p::debug_env(name); } }

// envs resident per device (live v2p_env batches of this process): what kernel_build = 0 decides by, launch by launch
static std::atomic<int64_t> g_resident_envs[64];
static int64_t resident_envs(int device) { return (device >= 0 && device < 64) ? g_resident_envs[device].load(std::memory_order_relaxed) : 0; }
static constexpr int64_t REGS_BUILD_MAX_ENVS = 5120;  // measured crossover of the two builds (profiles/r04e_dual_build.txt)

namespace v2p {
// out[N][525] + ws: only the env-per-lane cross-check schedule stages through global memory
int ensure_env_per_lane_buffers(v2p_env* e) {
    if (e->out && e->ws) return V2P_OK;
    const size_t N = (size_t)e->n;
    int rc = V2P_OK;
    if (!e->out) rc = check_hip(hipMalloc((void**)&e->out, sizeof(float) * OUT_SLOTS * N), "hipMalloc(out)");
    if (rc == V2P_OK && !e->ws) rc = check_hip(hipMalloc((void**)&e->ws, sizeof(float) * (size_t)physics_ws_slots() * N), "hipMalloc(ws)");
    if (rc == V2P_OK) rc = check_hip(hipMemset(e->out, 0, sizeof(float) * OUT_SLOTS * N), "hipMemset(out)");
    if (rc == V2P_OK) rc = check_hip(hipMemset(e->ws, 0, sizeof(float) * (size_t)physics_ws_slots() * N), "hipMemset(ws)");
    if (rc == V2P_OK) rc = check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize(env_per_lane buffers)");
    return rc;
}
}  // namespace v2p

static void profile_free(v2p_env* e);

extern "C" {

const char* v2p_last_error(void) { return g_err; }
int v2p_abi_version(void) { return V2P_ABI_VERSION; }

int v2p_model_create(const v2p_model_desc* d, int device, v2p_model** out) {
    if (!d || !out) { set_error("v2p_model_create: null argument"); return V2P_ERR_INVALID; }
    if (d->num_bodies != NB) { set_error("v2p_model_create: num_bodies must be %d, got %d", NB, d->num_bodies); return V2P_ERR_UNSUPPORTED; }
    if (d->hull_offsets[NB] > MAX_HULL_VERTS) { set_error("v2p_model_create: %d hull vertices exceed the limit %d", d->hull_offsets[NB], MAX_HULL_VERTS); return V2P_ERR_UNSUPPORTED; }
    v2p_model* m = new (std::nothrow) v2p_model();
    if (!m) { set_error("v2p_model_create: out of host memory"); return V2P_ERR_NOMEM; }
    memset(&m->host, 0, sizeof(m->host));
    DevModel& h = m->host;
    for (int b = 0; b < NB; ++b) {
        int p = d->parents[b];
        if ((b == 0 && p != -1) || (b > 0 && (p < 0 || p >= b))) {
            set_error("v2p_model_create: parents must be topologically ordered with a single root (body %d has parent %d)", b, p);
            delete m;
            return V2P_ERR_INVALID;
        }
        h.parents[b] = p;
        h.depth[b] = b == 0 ? 0 : h.depth[p] + 1;
        if (h.depth[b] >= MAX_DEPTH) { set_error("v2p_model_create: tree depth exceeds %d", MAX_DEPTH); delete m; return V2P_ERR_UNSUPPORTED; }
        for (int k = 0; k < 3; ++k) { h.shape.local_pos[b][k] = d->local_pos[3 * b + k]; h.shape.com[b][k] = d->com[3 * b + k]; }
        h.shape.mass[b] = d->mass[b];
        const float* I = d->inertia + 9 * b;
        h.shape.inertia[b][0] = I[0]; h.shape.inertia[b][1] = 0.5f * (I[1] + I[3]); h.shape.inertia[b][2] = 0.5f * (I[2] + I[6]);
        h.shape.inertia[b][3] = I[4]; h.shape.inertia[b][4] = 0.5f * (I[5] + I[7]); h.shape.inertia[b][5] = I[8];
        if (b > 0) {
            const float *kp = d->kp + 3 * (b - 1), *kd = d->kd + 3 * (b - 1), *ar = d->armature + 3 * (b - 1);
            if (kp[0] != kp[1] || kp[0] != kp[2] || kd[0] != kd[1] || kd[0] != kd[2] || ar[0] != ar[1] || ar[0] != ar[2]) {
                set_error("v2p_model_create: joint of body %d has per-axis gains; only isotropic spherical-joint gains are built", b);
                delete m;
                return V2P_ERR_UNSUPPORTED;
            }
            h.shape.kp[b] = kp[0]; h.shape.kd[b] = kd[0]; h.shape.arm[b] = ar[0];
        }
        for (int k = 0; k < 3; ++k) {
            h.shape.limit_lo[b][k] = b > 0 && d->limit_lower ? d->limit_lower[3 * (b - 1) + k] : -3.14159265f;
            h.shape.limit_hi[b][k] = b > 0 && d->limit_upper ? d->limit_upper[3 * (b - 1) + k] : 3.14159265f;
            if (!(h.shape.limit_lo[b][k] <= h.shape.limit_hi[b][k])) { set_error("v2p_model_create: body %d: joint range is empty", b); delete m; return V2P_ERR_INVALID; }
        }
    }
    {
        int off = 0;
        for (int b = 0; b < NB; ++b) {
            int n = d->hull_offsets[b + 1] - d->hull_offsets[b];
            int np = (n + HULL_PAD - 1) / HULL_PAD * HULL_PAD;
            if (n < 1 || n > 64) { set_error("v2p_model_create: body %d has %d hull vertices (1..64 supported)", b, n); delete m; return V2P_ERR_UNSUPPORTED; }
            if (off + np > MAX_HULL_VERTS) { set_error("v2p_model_create: padded hull vertices exceed the limit %d", MAX_HULL_VERTS); delete m; return V2P_ERR_UNSUPPORTED; }
            h.shape.hull_offsets[b] = off;
            h.shape.hull_count[b] = n;
            float r2 = 0.f, lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
            for (int v = 0; v < np; ++v) {
                const float* src = d->hull_verts + 3 * (d->hull_offsets[b] + (v < n ? v : n - 1));
                for (int k = 0; k < 3; ++k) {
                    h.shape.hull_verts[off + v][k] = src[k];
                    lo[k] = src[k] < lo[k] ? src[k] : lo[k];
                    hi[k] = src[k] > hi[k] ? src[k] : hi[k];
                }
                float n2 = src[0] * src[0] + src[1] * src[1] + src[2] * src[2];
                if (n2 > r2) r2 = n2;
            }
            h.shape.bound_radius[b] = sqrtf(r2);
            for (int k = 0; k < 3; ++k) { h.shape.aabb_c[b][k] = 0.5f * (lo[k] + hi[k]); h.shape.aabb_e[b][k] = 0.5f * (hi[k] - lo[k]) * 1.0001f + 1e-6f; }
            off += np;
        }
        h.shape.hull_offsets[NB] = off;
        h.shape.hull_cofs[0] = 0;
        for (int b = 0; b < NB; ++b) h.shape.hull_cofs[b + 1] = h.shape.hull_cofs[b] + h.shape.hull_count[b];
    }
    {
        int k = 0;
        for (int dpt = 0; dpt < MAX_DEPTH; ++dpt)
            for (int b = 0; b < NB; ++b)
                if (h.depth[b] == dpt) h.order[k++] = b;
    }
    {
        int nch[NB] = {0};
        h.max_depth = 0;
        h.multi_child_levels = 0;
        h.nonchain_levels = 0;
        h.max_hull_count = 0;
        for (int b = 0; b < NB; ++b) {
            for (int k = 0; k < 3; ++k) h.children[b][k] = -1;
            h.anc_mask[b] = 1 << b;
            if (h.depth[b] > h.max_depth) h.max_depth = h.depth[b];
            if (h.shape.hull_count[b] > h.max_hull_count) h.max_hull_count = h.shape.hull_count[b];
        }
        for (int b = 1; b < NB; ++b) {
            int p = h.parents[b];
            if (nch[p] >= 3) { set_error("v2p_model_create: link %d has more than 3 children", p); delete m; return V2P_ERR_UNSUPPORTED; }
            h.children[p][nch[p]++] = b;
            if (nch[p] > 1) h.multi_child_levels |= 1 << h.depth[b];
            h.anc_mask[b] |= h.anc_mask[p];
            if (p != b - 1) h.nonchain_levels |= 1 << h.depth[b];
            if (h.children[p][0] != p + 1) { set_error("v2p_model_create: links must be in depth-first order (first child of %d is %d)", p, h.children[p][0]); delete m; return V2P_ERR_UNSUPPORTED; }
        }
    }
    if (h.max_depth > 15) { set_error("v2p_model_create: tree deeper than 15 levels"); delete m; return V2P_ERR_UNSUPPORTED; }
    h.jump_rounds = 0;
    while ((1 << h.jump_rounds) <= h.max_depth) ++h.jump_rounds;
    for (int b = 0; b < NB; ++b) {
        h.anc_jump[b] = 0;
        for (int k = 0; k < 4; ++k) {
            int a = b, steps = 1 << k;
            while (steps > 0 && a > 0) { a = h.parents[a]; --steps; }
            h.anc_jump[b] = (int32_t)((uint32_t)h.anc_jump[b] | ((uint32_t)((steps == 0 && b != 0) ? a : 255) << (8 * k)));
        }
    }
    h.side_depths[0] = 0;
    for (int b = 1; b < NB; ++b) h.side_depths[b] = h.side_depths[h.parents[b]] | ((h.parents[b] != b - 1) ? 1 << h.depth[b] : 0);
    for (int b = 0; b < NB; ++b) {
        h.desc_mask[b] = 0;
        for (int j = 0; j < NB; ++j)
            if ((h.anc_mask[j] >> b) & 1) h.desc_mask[b] |= 1 << j;
    }
    {
        int nchild[NB] = {0}, nslot = 1;
        for (int b = 1; b < NB; ++b) nchild[h.parents[b]]++;
        for (int b = 0; b < NB; ++b) h.lam_slot[b] = -1;
        h.lam_slot[0] = 0;
        for (int b = 1; b < NB; ++b) {
            // a child that does not directly follow its parent needs the parent's Lambda from a saved slot
            int p = h.parents[b];
            if (p != b - 1 && h.lam_slot[p] < 0) {
                if (nslot >= MAX_BRANCH) { set_error("v2p_model_create: more than %d branching links", MAX_BRANCH); delete m; return V2P_ERR_UNSUPPORTED; }
                h.lam_slot[p] = nslot++;
            }
        }
        (void)nchild;
    }
    m->device = device;
    m->dev = nullptr;
    DeviceGuard g(device);
    if (!g.ok) { set_error("v2p_model_create: cannot select device %d", device); delete m; return V2P_ERR_HIP; }
    int rc = check_hip(hipMalloc((void**)&m->dev, sizeof(DevModel)), "hipMalloc(model)");
    if (rc == V2P_OK) rc = check_hip(hipMemcpy(m->dev, &m->host, sizeof(DevModel), hipMemcpyHostToDevice), "hipMemcpy(model)");
    if (rc != V2P_OK) { if (m->dev) (void)hipFree(m->dev); delete m; return rc; }
    *out = m;
    return V2P_OK;
}

void v2p_model_destroy(v2p_model* m) {
    if (!m) return;
    DeviceGuard g(m->device);
    if (m->dev) (void)hipFree(m->dev);
    delete m;
}

int v2p_mlib_create(const v2p_motion_tables* t, int device, v2p_mlib** out) {
    if (!t || !out) { set_error("v2p_mlib_create: null argument"); return V2P_ERR_INVALID; }
    if (t->num_motions <= 0 || t->num_frames_total <= 0) { set_error("v2p_mlib_create: empty motion library"); return V2P_ERR_INVALID; }
    const void* ptrs[] = {t->gts, t->grs, t->lrs, t->grvs, t->gravs, t->dvs, t->motion_lengths, t->motion_num_frames, t->motion_dt,
                          t->motion_min_verts_h, t->length_starts, t->motion_bodies};
    for (const void* p : ptrs)
        if (!p) { set_error("v2p_mlib_create: null table pointer"); return V2P_ERR_INVALID; }
    if (((uintptr_t)t->grs | (uintptr_t)t->lrs) & 15) { set_error("v2p_mlib_create: grs/lrs must be 16-byte aligned"); return V2P_ERR_INVALID; }
    for (int k = 0; k < 4; ++k)
        if (t->key_body_ids[k] < 0 || t->key_body_ids[k] >= NB) { set_error("v2p_mlib_create: key body id out of range"); return V2P_ERR_INVALID; }
    v2p_mlib* m = new (std::nothrow) v2p_mlib();
    if (!m) { set_error("v2p_mlib_create: out of host memory"); return V2P_ERR_NOMEM; }
    m->t = *t;
    m->device = device;
    *out = m;
    return V2P_OK;
}

void v2p_mlib_destroy(v2p_mlib* m) { delete m; }

int v2p_motion_state(const v2p_mlib* m, const int64_t* ids, const float* times, int64_t q, int adjust_height, float ground_tol,
                     float* const out[9], void* stream) {
    if (!m || !out || q < 0 || (q > 0 && (!ids || !times))) { set_error("v2p_motion_state: bad argument"); return V2P_ERR_INVALID; }
    DeviceGuard g(m->device);
    return launch_motion_state(m->t, ids, times, q, adjust_height, ground_tol, out, (hipStream_t)stream);
}

int v2p_reward(int64_t n, const float* body_pos, const float* body_rot, const float* tgt_pos, const float* tgt_rot, const float* dof_pos,
               const float* dof_vel, const float* tgt_dof_pos, const float* tgt_dof_vel, const float* w, const float specs[8], float* reward,
               float* sub, void* stream) {
    if (n < 0 || !w || !specs) { set_error("v2p_reward: bad argument"); return V2P_ERR_INVALID; }
    return launch_reward(n, body_pos, body_rot, tgt_pos, tgt_rot, dof_pos, dof_vel, tgt_dof_pos, tgt_dof_vel, w, specs, reward, sub,
                         (hipStream_t)stream);
}

int v2p_reset_flags(int64_t n, const int64_t* progress, const float* rb_pos, const float* heights, const float* cur_time,
                    const float* clip_len, float max_len, int early, int64_t* reset_out, int64_t* term_out, void* stream) {
    if (n < 0 || !heights) { set_error("v2p_reset_flags: bad argument"); return V2P_ERR_INVALID; }
    return launch_reset_flags(n, progress, rb_pos, heights, cur_time, clip_len, max_len, early, reset_out, term_out, (hipStream_t)stream);
}

int v2p_obs_imitation(int64_t n, const float* body_pos, const float* body_rot, const float* tgt_pos, const float* tgt_rot,
                      const float* dof_pos, const float* dof_vel, const float* tgt_dof_pos, const float* body_vel,
                      const float* body_ang_vel, const float* motion_bodies, const float* norm_mean, const float* norm_std, float norm_clip,
                      float* obs, void* stream) {
    if (n < 0 || ((norm_mean == nullptr) != (norm_std == nullptr))) { set_error("v2p_obs_imitation: bad argument"); return V2P_ERR_INVALID; }
    return launch_obs_imitation(n, body_pos, body_rot, tgt_pos, tgt_rot, dof_pos, dof_vel, tgt_dof_pos, body_vel, body_ang_vel,
                                motion_bodies, norm_mean, norm_std, norm_clip, obs, (hipStream_t)stream);
}

int v2p_obs_imitation_packed(int64_t rows, int64_t steps, const float* obs, const float* context_feat, int64_t ctx_frames, int64_t first_frame,
                             const float* norm_mean, const float* norm_std, float norm_clip, float* out, void* stream) {
    if (rows < 0 || steps < 1 || rows % steps || !obs || !context_feat || !out || first_frame < 0 || first_frame + steps > ctx_frames ||
        ((norm_mean == nullptr) != (norm_std == nullptr))) {
        set_error("v2p_obs_imitation_packed: bad argument");
        return V2P_ERR_INVALID;
    }
    return launch_obs_imitation_packed(rows, steps, obs, context_feat, ctx_frames, first_frame, norm_mean, norm_std, norm_clip, out, (hipStream_t)stream);
}

int v2p_policy_head(int64_t n, float* mu, const float* context_feat, int64_t ctx_frames, int64_t frame, const float* logstd, const float* noise,
                    float* action, float* sigma, float* neglogp, void* stream) {
    if (n < 0 || !mu || !context_feat || !logstd || !noise || !action || !neglogp || frame < 0 || frame >= ctx_frames) {
        set_error("v2p_policy_head: bad argument");
        return V2P_ERR_INVALID;
    }
    return launch_policy_head(n, mu, context_feat, ctx_frames, frame, logstd, noise, action, sigma, neglogp, (hipStream_t)stream);
}

int v2p_policy_head_record(int64_t n, float* mu, const float* context_feat, int64_t ctx_frames, int64_t frame, const float* logstd, const float* noise,
                           float* action, float* sigma_row, float* neglogp_row, float* action_row, float* mu_row, void* stream) {
    if (n < 0 || !mu || !context_feat || !logstd || !noise || !action || !neglogp_row || frame < 0 || frame >= ctx_frames) {
        set_error("v2p_policy_head_record: bad argument");
        return V2P_ERR_INVALID;
    }
    return launch_policy_head(n, mu, context_feat, ctx_frames, frame, logstd, noise, action, sigma_row, neglogp_row, (hipStream_t)stream, action_row, mu_row);
}

int v2p_gae(int64_t horizon, int64_t n, const float* fdones, const float* values, const float* rewards, const float* next_values, float gamma,
            float tau, float* advs, void* stream) {
    if (horizon < 0 || n < 0) { set_error("v2p_gae: bad argument"); return V2P_ERR_INVALID; }
    return launch_gae(horizon, n, fdones, values, rewards, next_values, gamma, tau, advs, (hipStream_t)stream);
}

int v2p_value_record(int64_t n, const float* value_raw, const double* running_mean, const double* running_var, float epsilon, const float* terminated,
                     float* values_row, float* next_values_row, void* stream) {
    if (n < 0 || (n > 0 && (!value_raw || ((running_mean == nullptr) != (running_var == nullptr)) || (next_values_row && !terminated)))) {
        set_error("v2p_value_record: bad argument");
        return V2P_ERR_INVALID;
    }
    return launch_value_record(n, value_raw, running_mean, running_var, epsilon, terminated, values_row, next_values_row, (hipStream_t)stream);
}

int v2p_rollout_record(int64_t n, const float* obs, int64_t obs_dim, const float* rew, const int64_t* reset, const int64_t* terminate, const float* sub_rewards,
                       float* next_obs_row, float* rewards_row, float* dones_row, float* dones, float* terminated, float* prev_dones, float* cur_rewards,
                       float* cur_lengths, double* acc, double* sub_acc, void* stream) {
    if (n < 0 || obs_dim < 0 || (n > 0 && (!rew || !reset || !terminate || !sub_rewards || !rewards_row || !dones_row || !dones || !terminated || !prev_dones ||
                                           !cur_rewards || !cur_lengths || !acc || !sub_acc || (next_obs_row && !obs)))) {
        set_error("v2p_rollout_record: bad argument");
        return V2P_ERR_INVALID;
    }
    return launch_rollout_record(n, obs, obs_dim, rew, reset, terminate, sub_rewards, next_obs_row, rewards_row, dones_row, dones, terminated, prev_dones, cur_rewards,
                                 cur_lengths, acc, sub_acc, (hipStream_t)stream);
}

int v2p_motion_tables_build(int64_t num_frames_total, int64_t num_clips, const double* local_rot, const double* root_trans, const int32_t* frame_clip,
                            const int64_t* clip_start, const int32_t* clip_frames, const double* clip_dt, const int32_t* parents, const double* local_pos,
                            int32_t per_clip_skeleton, float* gts, float* grs, float* lrs, float* grvs, float* gravs, float* dvs, void* stream) {
    if (num_frames_total < 0 || num_clips < 0 || !parents) { set_error("v2p_motion_tables_build: bad argument"); return V2P_ERR_INVALID; }
    if (num_frames_total > 0 && (!local_rot || !root_trans || !frame_clip || !clip_start || !clip_frames || !clip_dt || !local_pos || !gts || !grs || !lrs || !grvs || !gravs || !dvs)) {
        set_error("v2p_motion_tables_build: null buffer");
        return V2P_ERR_INVALID;
    }
    return launch_motion_tables_build(num_frames_total, local_rot, root_trans, frame_clip, clip_start, clip_frames, clip_dt, parents, local_pos, per_clip_skeleton ? 1 : 0,
                                      gts, grs, lrs, grvs, gravs, dvs, (hipStream_t)stream);
}

int v2p_shapes_compile(int32_t num_jobs, const double* points, const int32_t* job_offsets, int32_t max_points, const double* dirs, const int32_t* dir_offsets,
                       int32_t num_dir_tables, double density, int32_t max_verts, double eps_rel, double* mass, double* com, double* inertia, int32_t* num_verts,
                       int32_t* vert_ids, double* verts, int32_t* status, void* stream) {
    if (num_jobs < 0 || max_points < 0 || max_verts < 4 || max_verts > 64 || num_dir_tables < 1 || !(density > 0.0) || !(eps_rel >= 0.0)) { set_error("v2p_shapes_compile: bad argument"); return V2P_ERR_INVALID; }
    if (num_jobs > 0 && (!points || !job_offsets || !dirs || !dir_offsets || !mass || !com || !inertia || !num_verts || !vert_ids || !verts || !status)) {
        set_error("v2p_shapes_compile: null buffer");
        return V2P_ERR_INVALID;
    }
    return launch_shape_compile(num_jobs, points, job_offsets, max_points, dirs, dir_offsets, num_dir_tables, density, max_verts, eps_rel, mass, com, inertia,
                                num_verts, vert_ids, verts, status, (hipStream_t)stream);
}

static int env_create_impl(const v2p_model* const* shapes, int32_t num_shapes, const int32_t* env_shape_id, const v2p_mlib* mlib,
                           const v2p_sim_cfg* c, const int64_t* env_motion_id, int64_t n, const v2p_env_buffers* b, int device, v2p_env** out) {
    if (!shapes || num_shapes < 1 || !shapes[0]) { set_error("v2p_env_create: bad argument"); return V2P_ERR_INVALID; }
    const v2p_model* model = shapes[0];
    for (int32_t k = 1; k < num_shapes; ++k) {
        if (!shapes[k] || shapes[k]->device != device) { set_error("v2p_env_create_shapes: shape %d is null or lives on another device", k); return V2P_ERR_INVALID; }
        if (memcmp(shapes[k]->host.parents, model->host.parents, sizeof(model->host.parents))) {
            set_error("v2p_env_create_shapes: shape %d has a different body tree", k);
            return V2P_ERR_UNSUPPORTED;
        }
    }
    if (num_shapes > 1) {
        if (!env_shape_id) { set_error("v2p_env_create_shapes: env_shape_id is null"); return V2P_ERR_INVALID; }
        for (int64_t i = 0; i < n; ++i)
            if (env_shape_id[i] < 0 || env_shape_id[i] >= num_shapes) { set_error("v2p_env_create_shapes: env %lld has shape id %d", (long long)i, env_shape_id[i]); return V2P_ERR_INVALID; }
    }
    if (!model || !mlib || !c || !env_motion_id || !b || !out || n <= 0) { set_error("v2p_env_create: bad argument"); return V2P_ERR_INVALID; }
    if (model->device != device || mlib->device != device) { set_error("v2p_env_create: model/motion-lib live on another device"); return V2P_ERR_INVALID; }
    const void* req[] = {b->root_states, b->dof_state, b->rb_state, b->contact_force, b->dof_force, b->pd_target, b->obs, b->rew,
                         b->sub_rewards, b->reset, b->terminate, b->progress, b->cur_time, b->reset_time, b->target[0], b->target[1]};
    for (const void* p : req)
        if (!p) { set_error("v2p_env_create: a required buffer is null"); return V2P_ERR_INVALID; }
    if (c->substeps < 1 || c->control_freq_inv < 1 || c->sim_dt <= 0.f || c->num_solver_iterations < 0 ||
        c->residual_hold_sims < 0 || c->residual_hold_sims > c->control_freq_inv) {
        set_error("v2p_env_create: bad sim parameters");
        return V2P_ERR_INVALID;
    }
    v2p_env* e = new (std::nothrow) v2p_env();
    if (!e) { set_error("v2p_env_create: out of host memory"); return V2P_ERR_NOMEM; }
    memset(e, 0, sizeof(*e));
    e->model = model;
    e->mlib = mlib;
    e->buf = *b;
    e->n = n;
    e->device = device;
    e->motion_id = env_motion_id;
    if (c->schedule != 0 && c->schedule != 1) { set_error("v2p_env_create: schedule must be 0 or 1"); delete e; return V2P_ERR_INVALID; }
    if (c->solver_type != 0 && c->solver_type != 1) { set_error("v2p_env_create: solver_type must be 0 (PGS) or 1 (TGS)"); delete e; return V2P_ERR_INVALID; }
    if (c->solver_type == 1 && c->schedule == 1) { set_error("v2p_env_create: the env-per-lane cross-check kernel solves PGS only"); delete e; return V2P_ERR_UNSUPPORTED; }
    if (c->kernel_build < 0 || c->kernel_build > 2) { set_error("v2p_env_create: kernel_build must be 0 (engine's choice), 1 (LDS-parked) or 2 (registers)"); delete e; return V2P_ERR_INVALID; }
    if (c->num_velocity_iterations != 0) {
        set_error("v2p_env_create: sim.physx.num_velocity_iterations = %d: the engine's contact solvers have no separate velocity pass (the reference's configs use 0)", c->num_velocity_iterations);
        delete e;
        return V2P_ERR_UNSUPPORTED;
    }
    if (!(c->bounce_threshold_velocity >= 0.f) || (c->enable_contact && !(c->rest_offset < c->contact_offset))) {
        set_error("v2p_env_create: bounce_threshold_velocity must be >= 0 and (with contacts on) rest_offset below contact_offset");
        delete e;
        return V2P_ERR_INVALID;
    }
    if (c->friction_frame != 0 && c->friction_frame != 1) { set_error("v2p_env_create: friction_frame must be 0 (world) or 1 (velocity)"); delete e; return V2P_ERR_INVALID; }
    if (c->friction_frame == 1 && c->schedule == 1) { set_error("v2p_env_create: the env-per-lane cross-check kernel solves in the world friction frame only"); delete e; return V2P_ERR_UNSUPPORTED; }
    if (c->joint_limits && (c->schedule == 1 || !c->enable_contact)) {
        set_error("v2p_env_create: joint_limits needs the link-per-lane schedule and contacts on");
        delete e;
        return V2P_ERR_UNSUPPORTED;
    }
    e->schedule = c->schedule;
    EnvParams& p = e->p;
    p.h = c->sim_dt / (float)c->substeps;
    p.nsub = c->substeps * c->control_freq_inv;
    e->substeps_per_sim = c->substeps;
    p.hold_sub = c->residual_hold_sims * c->substeps;
    p.n_iter = c->num_solver_iterations;
    p.enable_contact = c->enable_contact;
    p.gravity_z = c->gravity_z; p.mu = c->friction; p.contact_offset = c->contact_offset; p.max_depen = c->max_depenetration_velocity;
    p.erp = c->erp; p.ang_damp = c->angular_damping; p.max_ang_vel = c->max_angular_velocity;
    p.pd_tar_lim = c->pd_tar_lim; p.res_force_scale = c->residual_force_scale; p.res_torque_scale = c->residual_torque_scale;
    p.ground_tolerance = c->ground_tolerance; p.max_episode_length = c->max_episode_length;
    p.enable_early_termination = c->enable_early_termination;
    p.freeze_terminated = c->freeze_terminated_envs;
    p.solver_type = c->solver_type;
    p.joint_limits = c->joint_limits ? 1 : 0;
    p.limit_margin = c->limit_margin <= 0.f ? 0.05f : c->limit_margin;  // (a zero-initialised cfg gets the default)
    p.rest_offset = c->rest_offset;
    p.friction_frame = c->friction_frame;
    p.bounce_threshold = c->bounce_threshold_velocity;
    p.context_length = c->context_length; p.context_padding = c->context_padding;
    p.dt = (float)c->control_freq_inv * c->sim_dt;
    memcpy(p.term_heights, c->term_heights, sizeof(p.term_heights));
    memcpy(p.body_pos_weights, c->body_pos_weights, sizeof(p.body_pos_weights));
    memcpy(p.reward_specs, c->reward_specs, sizeof(p.reward_specs));
    for (int b = 0; b < NB; ++b) p.aug[b] = b ? model->host.shape.arm[b] + p.h * model->host.shape.kd[b] + p.h * p.h * model->host.shape.kp[b] : 0.f;
    e->num_shapes = num_shapes;
    DeviceGuard g(device);
    if (!g.ok) { set_error("v2p_env_create: cannot select device %d", device); delete e; return V2P_ERR_HIP; }
    size_t N = (size_t)n;
    int rc = check_hip(hipMalloc((void**)&e->state, sizeof(float) * STATE_SLOTS * N), "hipMalloc(state)");
    if (rc == V2P_OK) rc = check_hip(hipMalloc((void**)&e->ctrl, sizeof(float) * CTRL_SLOTS * N), "hipMalloc(ctrl)");
    if (rc == V2P_OK && c->debug_contacts >= 1) rc = check_hip(hipMalloc((void**)&e->contact_ids, sizeof(int32_t) * NB * 4 * N), "hipMalloc(contact_ids)");
    if (rc == V2P_OK) rc = check_hip(hipMemset(e->state, 0, sizeof(float) * STATE_SLOTS * N), "hipMemset(state)");
    if (rc == V2P_OK) rc = check_hip(hipMemset(e->ctrl, 0, sizeof(float) * CTRL_SLOTS * N), "hipMemset(ctrl)");
    if (rc == V2P_OK && e->schedule == 1) rc = ensure_env_per_lane_buffers(e);
    if (rc == V2P_OK && c->debug_contacts >= 2) {
        rc = check_hip(hipMalloc((void**)&e->contact_ids_sub, sizeof(int32_t) * NB * 4 * N * (size_t)p.nsub), "hipMalloc(contact_ids_sub)");
        if (rc == V2P_OK) rc = check_hip(hipMemset(e->contact_ids_sub, 0xff, sizeof(int32_t) * NB * 4 * N * (size_t)p.nsub), "hipMemset(contact_ids_sub)");
    }
    if (rc == V2P_OK && e->contact_ids) rc = check_hip(hipMemset(e->contact_ids, 0xff, sizeof(int32_t) * NB * 4 * N), "hipMemset(contact_ids)");
    if (rc == V2P_OK && num_shapes > 1) {
        // per-env body shapes: the numeric tables of every shape + each shape's joint-diagonal augmentation, indexed by env_shape
        std::vector<float> aug((size_t)num_shapes * NB, 0.f);
        rc = check_hip(hipMalloc((void**)&e->shapes_dev, sizeof(DevShape) * (size_t)num_shapes), "hipMalloc(shapes)");
        for (int32_t k = 0; k < num_shapes && rc == V2P_OK; ++k) {
            const DevShape& sh = shapes[k]->host.shape;
            rc = check_hip(hipMemcpy(e->shapes_dev + k, &sh, sizeof(DevShape), hipMemcpyHostToDevice), "hipMemcpy(shape)");
            for (int bb = 1; bb < NB; ++bb) aug[(size_t)k * NB + bb] = sh.arm[bb] + p.h * sh.kd[bb] + p.h * p.h * sh.kp[bb];
        }
        if (rc == V2P_OK) rc = check_hip(hipMalloc((void**)&e->shape_aug_dev, sizeof(float) * aug.size()), "hipMalloc(shape_aug)");
        if (rc == V2P_OK) rc = check_hip(hipMemcpy(e->shape_aug_dev, aug.data(), sizeof(float) * aug.size(), hipMemcpyHostToDevice), "hipMemcpy(shape_aug)");
        if (rc == V2P_OK) rc = check_hip(hipMalloc((void**)&e->env_shape_dev, sizeof(int32_t) * N), "hipMalloc(env_shape)");
        if (rc == V2P_OK) rc = check_hip(hipMemcpy(e->env_shape_dev, env_shape_id, sizeof(int32_t) * N, hipMemcpyHostToDevice), "hipMemcpy(env_shape)");
        if (rc == V2P_OK) e->schedule = 0;  // the env-per-lane cross-check kernel is single-shape
    }
    e->pair_period = c->pair_envs_by_load ? 1 : 0;
    e->substep_jobs = c->substep_jobs ? 1 : 0;
    {   // 1 = the engine decides launch by launch: cutting pays once the env pairs no longer fit the GPU's wave slots in one round
        // (CUs x 4 SIMDs x 3 waves; measured: at <= 2/3 of the slots whole control steps per workgroup are 0.3 ... 8 % faster); 2 = always
        hipDeviceProp_t prop;
        e->job_min_blocks = 0;
        const bool have = hipGetDeviceProperties(&prop, device) == hipSuccess;
        if (c->substep_jobs == 1 && have) e->job_min_blocks = prop.multiProcessorCount * 8;
        e->job_len2_blocks = have ? prop.multiProcessorCount * 32 : 8192;
    }
    // (mixing trades total work for a shorter critical path: it pays while the launch is as long as its heaviest pair, i.e. up to
    // ~4 env pairs per wave slot; beyond that the launch is throughput bound and pairs of equals are cheaper)
    // (the kernels with joint-limit rows or a ball run 2 waves per SIMD: there pairs of equals measured best, profiles/r02g_racket_ball_sweep.txt)
    // which build of the link-per-lane kernel this batch runs (see the head of this file): v2p_sim_cfg.kernel_build, 0 = by the number of
    // envs RESIDENT on the device - the batches of a process that share a GPU (rollout groups) are bound by instruction issue together,
    // whatever the size of each - re-evaluated launch by launch (choose_build); the value here is the one a lone batch would get
    e->kernel_build = c->kernel_build;
    e->ll_regs_build = c->kernel_build ? (c->kernel_build == 2) : (resident_envs(device) + n <= REGS_BUILD_MAX_ENVS);
    e->pair_mix_default = c->pair_mix_permille < 0 ? 1 : 0;
    // defaults: measured best.  Round 2 (profiles/r02_job_mono_sweep.txt): 250 / 250; re-swept on the round-4 kernel (profiles/r04_mono_mix_sweep.txt:
    // 5 x 4 grid at 8192 envs, then across TGS / djokovic / per-clip shapes / 4096 and 12288 envs): 60 / 150 is +1 .. 2 % everywhere - with the
    // walk the heaviest chains are shorter, fewer pairs need to keep their substeps in one workgroup
    // (the register build runs where a launch is as long as its heaviest wave: there every heavy env takes a light partner, 500 - +1.3 % at 1024
    // and 4096 envs against 150, profiles/r04e_dual_build.txt)
    e->pair_mix_permille = c->pair_mix_permille < 0 ? ((n <= 12288 && !c->joint_limits) ? (e->ll_regs_build ? 500 : 150) : 0) : c->pair_mix_permille;
    // (above 12288 envs, with joint limits or with a ball - where the heavy x light mix is off - 250 stays 0.2 .. 1 % better)
    e->job_mono_default = c->job_mono_permille < 0 ? 1 : 0;
    e->job_mono_permille = c->job_mono_permille < 0 ? ((n <= 12288 && !c->joint_limits) ? 60 : 250) : c->job_mono_permille;
    if (e->pair_mix_permille > 500 || e->job_mono_permille > 1000) { set_error("v2p_env_create: pair_mix_permille <= 500, job_mono_permille <= 1000"); v2p_env_destroy(e); return V2P_ERR_INVALID; }
    if (rc == V2P_OK && e->substep_jobs) {
        const size_t words = (size_t)v2p::job_wave_slots(N) + 2;
        rc = check_hip(hipMalloc((void**)&e->job_progress, sizeof(int32_t) * words), "hipMalloc(job_progress)");
        if (rc == V2P_OK) rc = check_hip(hipMemset(e->job_progress, 0, sizeof(int32_t) * words), "hipMemset(job_progress)");
        // the state as the jobs hand it over: 50 16-byte chunks per env (see physics_ll.hip)
        if (rc == V2P_OK) rc = check_hip(hipMalloc((void**)&e->job_hand, sizeof(float) * HAND_FLOATS * N * (size_t)(p.nsub > 1 ? p.nsub - 1 : 1)), "hipMalloc(job_hand)");
        // ~20 ms: far beyond the longest chain of substeps of a launch.  (job_timeout_spins < 0: tests force the recovery path)
        e->job_timeout_spins = c->job_timeout_spins == 0 ? 50000l : (c->job_timeout_spins < 0 ? 0l : (long)c->job_timeout_spins);
        // substeps per job: 1 while the launch is short of jobs, 2 once there are plenty (>= CUs x 32 env pairs: measured crossover at
        // 16384 envs - a job's prologue / hand-over is ~8 % of a one-substep job); v2p_sim_cfg.job_len: A/B switch
        e->job_len = c->job_len > 0 ? c->job_len : 0;
        e->job_lead = c->job_lead == 0 ? -1 : (c->job_lead < 0 ? 0 : c->job_lead);  // -1: the engine decides (see launch_env_physics_ll)
    }
    e->job_interleave = c->job_no_interleave ? 0 : 1;  // (A/B switch)
    if (rc == V2P_OK) rc = check_hip(hipMalloc((void**)&e->pair_key, sizeof(int32_t) * N), "hipMalloc(pair_key)");
    if (rc == V2P_OK) rc = check_hip(hipMalloc((void**)&e->pair_pos, sizeof(int32_t) * N), "hipMalloc(pair_pos)");
    if (rc == V2P_OK) rc = check_hip(hipMalloc((void**)&e->perm, sizeof(int32_t) * N), "hipMalloc(perm)");
    if (rc == V2P_OK) rc = check_hip(hipMalloc((void**)&e->pair_hist, sizeof(int32_t) * (4 * PAIR_BINS + 1)), "hipMalloc(pair_hist)");
    for (int k = 0; k < 2 && rc == V2P_OK; ++k) rc = check_hip(hipMalloc((void**)&e->pair_list[k], sizeof(int32_t) * PAIR_BINS * (size_t)N), "hipMalloc(pair_list)");
    if (rc == V2P_OK) rc = check_hip(hipMalloc((void**)&e->pair_slot_env, sizeof(int32_t) * N), "hipMalloc(pair_slot_env)");
    if (rc == V2P_OK) {
        e->pair_starts[0] = e->pair_hist + PAIR_BINS;
        e->pair_starts[1] = e->pair_hist + 2 * PAIR_BINS;
        e->pair_start = e->pair_starts[0];
        e->pair_done = e->pair_hist + 4 * PAIR_BINS;
        rc = check_hip(hipMemset(e->pair_hist, 0, sizeof(int32_t) * (4 * PAIR_BINS + 1)), "hipMemset(pair_hist)");
    }
    if (rc == V2P_OK) rc = check_hip(hipMemset(e->pair_key, 0, sizeof(int32_t) * N), "hipMemset(pair_key)");
    if (rc == V2P_OK) {
        std::vector<int32_t> iota(N);
        for (size_t i = 0; i < N; ++i) iota[i] = (int32_t)i;
        rc = check_hip(hipMemcpy(e->perm, iota.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice), "hipMemcpy(perm)");
        if (rc == V2P_OK) rc = check_hip(hipMemcpy(e->pair_pos, iota.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice), "hipMemcpy(pair_pos)");
    }
    if (rc == V2P_OK) rc = check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize(env_create)");
    if (rc == V2P_OK && debug_env("V2P_WAVE_TIMES")) {
        // (one record per wave; per JOB in a V2P_LL_TIMELINE build: up to nsub per wave)
        rc = check_hip(hipMalloc((void**)&e->wave_times, sizeof(long long) * 4 * (N / 2 + 1) * (size_t)p.nsub), "hipMalloc(wave_times)");
        if (rc == V2P_OK) rc = check_hip(hipMemset(e->wave_times, 0, sizeof(long long) * 4 * (N / 2 + 1) * (size_t)p.nsub), "hipMemset(wave_times)");
    }
    if (rc == V2P_OK && debug_env("V2P_PHASE_TIMING")) {
        rc = check_hip(hipMalloc((void**)&e->prof, sizeof(long long) * 24), "hipMalloc(prof)");
        if (rc == V2P_OK) rc = check_hip(hipMemset(e->prof, 0, sizeof(long long) * 24), "hipMemset(prof)");
    }
    if (rc != V2P_OK) { v2p_env_destroy(e); return rc; }
    if (device >= 0 && device < 64) { g_resident_envs[device] += n; e->counted_resident = 1; }
    *out = e;
    return V2P_OK;
}

int v2p_env_create(const v2p_model* model, const v2p_mlib* mlib, const v2p_sim_cfg* c, const int64_t* env_motion_id, int64_t n,
                   const v2p_env_buffers* b, int device, v2p_env** out) {
    return env_create_impl(&model, 1, nullptr, mlib, c, env_motion_id, n, b, device, out);
}

int v2p_env_create_shapes(const v2p_model* const* shapes, int32_t num_shapes, const int32_t* env_shape_id, const v2p_mlib* mlib,
                          const v2p_sim_cfg* c, const int64_t* env_motion_id, int64_t n, const v2p_env_buffers* b, int device, v2p_env** out) {
    return env_create_impl(shapes, num_shapes, env_shape_id, mlib, c, env_motion_id, n, b, device, out);
}

void v2p_env_destroy(v2p_env* e) {
    if (!e) return;
    if (e->counted_resident) g_resident_envs[e->device] -= e->n;
    DeviceGuard g(e->device);
    if (e->state) (void)hipFree(e->state);
    if (e->ctrl) (void)hipFree(e->ctrl);
    if (e->out) (void)hipFree(e->out);
    if (e->ws) (void)hipFree(e->ws);
    if (e->contact_ids) (void)hipFree(e->contact_ids);
    if (e->contact_ids_sub) (void)hipFree(e->contact_ids_sub);
    if (e->job_progress) (void)hipFree(e->job_progress);
    if (e->job_hand) (void)hipFree(e->job_hand);
    if (e->ball && e->ball->contact_part) (void)hipFree(e->ball->contact_part);
    delete e->ball;
    if (e->err_host) { (void)hipHostFree(e->err_host); (void)hipEventDestroy(e->err_event); }
    profile_free(e);
    if (e->shapes_dev) (void)hipFree(e->shapes_dev);
    if (e->shape_aug_dev) (void)hipFree(e->shape_aug_dev);
    if (e->env_shape_dev) (void)hipFree(e->env_shape_dev);
    if (e->pair_key) (void)hipFree(e->pair_key);
    if (e->pair_pos) (void)hipFree(e->pair_pos);
    if (e->pair_hist) (void)hipFree(e->pair_hist);
    if (e->perm) (void)hipFree(e->perm);
    for (int k = 0; k < 2; ++k) if (e->pair_list[k]) (void)hipFree(e->pair_list[k]);
    if (e->pair_slot_env) (void)hipFree(e->pair_slot_env);
    if (e->wave_times) {
        const size_t nw = ((size_t)e->n / 2 + 1) * (size_t)e->p.nsub;  // (records that were never written stay zero and are skipped)
        std::vector<long long> h(nw * 4);
        FILE* f = fopen(debug_env("V2P_WAVE_TIMES") ? debug_env("V2P_WAVE_TIMES") : "wave_times.bin", "wb");
        if (f && hipMemcpy(h.data(), e->wave_times, sizeof(long long) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) fwrite(h.data(), sizeof(long long), h.size(), f);
        if (f) fclose(f);
        (void)hipFree(e->wave_times);
    }
    if (e->prof) {
        long long h[24];
        if (hipMemcpy(h, e->prof, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr,
                    "[v2p phase cycles, workgroup 0] link-per-lane: counter k = phase k-1 of {pass1, pass2, root+pass3, contacts, lambda, sweep, "
                    "integrate}; env-per-lane: {stage, pass1, pass2, root+pass3, contacts, lambda, sweep, integrate}: "
                    "%lld %lld %lld %lld %lld %lld %lld %lld | block updates %lld touched-sum %lld substeps %lld | "
                    "sweep: rows %lld up %lld contact-rounds %lld down %lld manifold-reductions %lld | contacts: cull %lld rounds %lld points %lld | null updates %lld\n",
                    h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15], h[16], h[17], h[18], h[19]);
        (void)hipFree(e->prof);
    }
    delete e;
}

int v2p_env_reset(v2p_env* e, const int64_t* env_ids, int64_t n, const float* motion_times, void* stream) {
    if (!e || !motion_times || n < 0 || n > e->n) { set_error("v2p_env_reset: bad argument"); return V2P_ERR_INVALID; }
    DeviceGuard g(e->device);
    if (!env_ids || n == e->n) e->build_latched = 0;  // an epoch boundary: the next launch may choose its build anew (kernel_build 0)
    return launch_env_reset(e, env_ids, env_ids ? n : e->n, motion_times, (hipStream_t)stream);
}

int v2p_env_context(v2p_env* e, const int64_t* env_ids, int64_t n, const float* motion_times, void* stream) {
    if (!e || !motion_times || n < 0 || n > e->n) { set_error("v2p_env_context: bad argument"); return V2P_ERR_INVALID; }
    if (!e->buf.context_feat) { set_error("v2p_env_context: the env was created without a context buffer"); return V2P_ERR_INVALID; }
    DeviceGuard g(e->device);
    return launch_env_context(e, env_ids, env_ids ? n : e->n, motion_times, (hipStream_t)stream);
}

int v2p_env_pre_physics(v2p_env* e, float* actions, void* stream) {
    if (!e || !actions) { set_error("v2p_env_pre_physics: bad argument"); return V2P_ERR_INVALID; }
    DeviceGuard g(e->device);
    return launch_env_pre(e, actions, (hipStream_t)stream);
}

// kernel_build = 0: the build follows the envs resident on the device (a second rollout group created after this batch moves both to the
// three-wave build); the heavy x light pairing share follows the build where it was left to the engine.  The choice is LATCHED: taken at
// the first launch after the batch was created or reset as a whole (an epoch boundary: every env restarts from a reference state) and
// kept until the next such reset - the two builds agree to rounding only, so a live batch must not change build in the middle of an
// epoch because an unrelated batch (an eval task next to training) came or went (advisor r5).
static int build_wanted(const v2p_env* e) { return resident_envs(e->device) <= REGS_BUILD_MAX_ENVS ? 1 : 0; }
static void choose_build(v2p_env* e) {
    if (e->kernel_build != 0 || e->build_latched) return;
    e->build_latched = 1;
    const int regs = build_wanted(e);
    if (regs == e->ll_regs_build) return;
    e->ll_regs_build = regs;
    if (e->pair_mix_default && !e->ball) e->pair_mix_permille = (e->n <= 12288 && !e->p.joint_limits) ? (regs ? 500 : 150) : 0;
}

// the physics launch of either schedule, bracketed by events while a measurement is open
static int physics_launch(v2p_env* e, hipStream_t s, float* actions, int* fused_post = nullptr) {
    // (sampled: launch L of the measurement is bracketed when L % stride == (L / period) % stride - every position of a period-long
    // epoch is met once in `stride` epochs)
    bool rec = e->prof_ev && e->prof_n < e->prof_cap;
    if (e->prof_ev) {
        const int64_t L = e->prof_seen++;
        if (e->prof_stride > 1) rec = rec && (L % e->prof_stride) == (L / e->prof_period) % e->prof_stride;
    }
    choose_build(e);
    if (rec) (void)hipEventRecord(e->prof_ev[2 * e->prof_n], s);
    int rc = e->schedule != 0 ? launch_env_physics(e, s)
                              : (e->ll_regs_build ? v2p_regs::launch_env_physics_ll(e, s, actions, fused_post) : launch_env_physics_ll(e, s, actions, fused_post));
    if (rec) { (void)hipEventRecord(e->prof_ev[2 * e->prof_n + 1], s); ++e->prof_n; }
    return rc;
}

int v2p_env_physics(v2p_env* e, void* stream) {
    if (!e) { set_error("v2p_env_physics: bad argument"); return V2P_ERR_INVALID; }
    DeviceGuard g(e->device);
    if (e->schedule != 0) { int rc = ensure_env_per_lane_buffers(e); if (rc != V2P_OK) return rc; }
    return physics_launch(e, (hipStream_t)stream, nullptr);
}

int v2p_env_attach_ball(v2p_env* e, const v2p_ball_cfg* c, const v2p_ball_buffers* b) {
    if (!e || !c || !b) { set_error("v2p_env_attach_ball: null argument"); return V2P_ERR_INVALID; }
    if (!b->ball_state || !b->racket_state || !b->ball_per_sim || !b->racket_hit_per_sim || !b->ball_contact) { set_error("v2p_env_attach_ball: a buffer is null"); return V2P_ERR_INVALID; }
    if (c->racket_link < 1 || c->racket_link >= NB || c->num_cylinders < 0 || c->num_cylinders > 2 || !(c->radius > 0.f) || !(c->mass > 0.f) || !(c->inertia > 0.f)) {
        set_error("v2p_env_attach_ball: bad ball parameters");
        return V2P_ERR_INVALID;
    }
    {
        const int nflags = (b->has_bounce != nullptr) + (b->has_bounce_now != nullptr) + (b->bounce_pos != nullptr) + (b->has_racket_contact != nullptr) + (b->has_racket_contact_now != nullptr);
        if (nflags != 0 && nflags != 5) { set_error("v2p_env_attach_ball: give all five flag buffers or none"); return V2P_ERR_INVALID; }
    }
    if (e->schedule != 0 || !e->p.enable_contact) {
        set_error("v2p_env_attach_ball: racket + ball needs the link-per-lane schedule and contacts on");
        return V2P_ERR_UNSUPPORTED;
    }
    if (e->p.rest_offset != 0.f) {
        set_error("v2p_env_attach_ball: sim.physx.rest_offset != 0 is modelled for the hull x plane rows only, not for the ball's rows");
        return V2P_ERR_UNSUPPORTED;
    }
    if (!e->ball) e->ball = new (std::nothrow) BallDev();
    if (!e->ball) { set_error("v2p_env_attach_ball: out of host memory"); return V2P_ERR_NOMEM; }
    BallDev& d = *e->ball;
    d.radius = c->radius; d.mass = c->mass; d.inv_mass = 1.f / c->mass; d.inv_inertia = 1.f / c->inertia;
    d.rest_ground = c->restitution_ground; d.fric_ground = c->friction_ground; d.rest_racket = c->restitution_racket; d.fric_racket = c->friction_racket;
    d.bounce_thr = c->bounce_threshold_velocity; d.ang_damp = c->angular_damping; d.max_ang_vel = c->max_angular_velocity; d.spin_scale = c->spin_scale;
    d.racket_link = c->racket_link; d.ncyl = c->num_cylinders; d.enabled = 1;
    d.sub_per_sim = e->substeps_per_sim;
    memcpy(d.cyl, c->cylinders, sizeof(d.cyl));
    memcpy(d.racket_off, c->racket_offset, sizeof(d.racket_off));
    d.state = b->ball_state; d.racket_state = b->racket_state; d.per_sim = b->ball_per_sim; d.hit_per_sim = b->racket_hit_per_sim; d.contact = b->ball_contact;
    d.rest_body = c->restitution_body; d.fric_body = c->friction_body; d.body_contacts = c->body_contacts ? 1 : 0;
    d.bounce_height = c->bounce_height; d.poll_hits = c->poll_racket_hits ? 1 : 0;
    d.body_contact = b->ball_body_contact;
    d.has_bounce = b->has_bounce; d.has_bounce_now = b->has_bounce_now; d.bounce_pos = b->bounce_pos;
    d.has_hit = b->has_racket_contact; d.has_hit_now = b->has_racket_contact_now;
    d.contact_sum = b->contact_force_sum;
    if (d.contact_sum && !d.contact_part) {
        const size_t nsim = (size_t)(e->p.nsub / e->substeps_per_sim);
        int rc = check_hip(hipMalloc((void**)&d.contact_part, sizeof(float) * (size_t)e->n * nsim * NB * 3), "hipMalloc(contact_part)");
        if (rc != V2P_OK) return rc;
    }
    if (e->pair_mix_default) e->pair_mix_permille = 0;
    if (e->job_mono_default) e->job_mono_permille = 250;
    return V2P_OK;
}

int v2p_env_check(v2p_env* e, void* stream) {
    if (!e) { set_error("v2p_env_check: bad argument"); return V2P_ERR_INVALID; }
    DeviceGuard g(e->device);
    int rc = check_hip(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
    if (rc != V2P_OK || !e->job_progress) return rc;
    int32_t count[2] = {0, 0};
    rc = check_hip(hipMemcpy(count, e->job_progress + v2p::job_wave_slots(e->n), sizeof(count), hipMemcpyDeviceToHost), "hipMemcpy(job recovery counters)");
    if (rc == V2P_OK) { e->job_recoveries = count[0]; e->jobs_skipped = count[1]; }
    if (rc == V2P_OK && e->jobs_skipped > e->jobs_skipped_reported) {
        // a late job of a cut pair found its step complete and did not run: its substeps were replayed by its successors (results are the
        // same bits), but what only IT publishes - exposed PD targets, the in-place masking of dead envs' actions, the ball's per-call
        // records - is missing for that step
        set_error("v2p_env_check: %lld substep job(s) started after their env pair's step was complete and were skipped: the per-call records they own were not "
                  "published for those steps (dispatch far out of order; v2p_sim_cfg.substep_jobs = 0 avoids it)", (long long)(e->jobs_skipped - e->jobs_skipped_reported));
        e->jobs_skipped_reported = e->jobs_skipped;
        return V2P_ERR_INTERNAL;
    }
    return rc;
}

int v2p_env_check_async(v2p_env* e, void* stream) {
    if (!e) { set_error("v2p_env_check_async: bad argument"); return V2P_ERR_INVALID; }
    if (!e->job_progress) return V2P_OK;
    DeviceGuard g(e->device);
    int rc = V2P_OK;
    if (!e->err_host) {
        rc = check_hip(hipHostMalloc((void**)&e->err_host, 2 * sizeof(int32_t), hipHostMallocDefault), "hipHostMalloc(job recovery counter)");
        if (rc == V2P_OK) {
            rc = check_hip(hipEventCreateWithFlags(&e->err_event, hipEventDisableTiming), "hipEventCreate(job recovery counter)");
            if (rc != V2P_OK) { (void)hipHostFree(e->err_host); e->err_host = nullptr; }  // (no half-built pair: a later call starts over)
        } else {
            e->err_host = nullptr;
        }
        if (rc != V2P_OK) return rc;
        e->err_host[0] = e->err_host[1] = 0;
    } else if (e->err_pending && hipEventQuery(e->err_event) == hipSuccess) {
        e->err_pending = 0;
        e->job_recoveries = e->err_host[0];
        e->jobs_skipped = e->err_host[1];
    }
    if (!e->err_pending) {  // fetch the counter as it stands behind everything enqueued so far; looked at by the next call
        rc = check_hip(hipMemcpyAsync(e->err_host, e->job_progress + v2p::job_wave_slots(e->n), 2 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream),
                       "hipMemcpyAsync(job recovery counter)");
        if (rc == V2P_OK) rc = check_hip(hipEventRecord(e->err_event, (hipStream_t)stream), "hipEventRecord(job recovery counter)");
        if (rc == V2P_OK) e->err_pending = 1;
    }
    return rc;
}

int v2p_env_job_recoveries(v2p_env* e, int64_t* count) {
    if (!e || !count) { set_error("v2p_env_job_recoveries: bad argument"); return V2P_ERR_INVALID; }
    *count = e->job_recoveries;
    return V2P_OK;
}

int v2p_env_jobs_skipped(v2p_env* e, int64_t* count) {
    if (!e || !count) { set_error("v2p_env_jobs_skipped: bad argument"); return V2P_ERR_INVALID; }
    *count = e->jobs_skipped;
    return V2P_OK;
}

static void profile_free(v2p_env* e) {
    if (!e->prof_ev) return;
    for (int64_t i = 0; i < 2 * e->prof_cap; ++i)
        if (e->prof_ev[i]) (void)hipEventDestroy(e->prof_ev[i]);
    delete[] e->prof_ev;
    e->prof_ev = nullptr;
    e->prof_cap = e->prof_n = 0;
}

int v2p_env_profile_begin(v2p_env* e, int64_t max_launches) { return v2p_env_profile_begin_sampled(e, max_launches, 1, 1); }

int v2p_env_profile_begin_sampled(v2p_env* e, int64_t max_launches, int32_t stride, int32_t period) {
    if (!e || max_launches <= 0 || max_launches > (1 << 20) || stride < 1 || period < 1) { set_error("v2p_env_profile_begin: bad argument"); return V2P_ERR_INVALID; }
    DeviceGuard g(e->device);
    profile_free(e);
    e->prof_stride = stride;
    e->prof_period = period;
    e->prof_seen = 0;
    e->prof_ev = new (std::nothrow) hipEvent_t[2 * max_launches]();
    if (!e->prof_ev) { set_error("v2p_env_profile_begin: out of host memory"); return V2P_ERR_NOMEM; }
    e->prof_cap = max_launches;
    for (int64_t i = 0; i < 2 * max_launches; ++i) {
        int rc = check_hip(hipEventCreate(&e->prof_ev[i]), "hipEventCreate");
        if (rc != V2P_OK) { profile_free(e); return rc; }
    }
    return V2P_OK;
}

int v2p_env_profile_end(v2p_env* e, double* physics_ms_total, int64_t* launches) {
    if (!e || !physics_ms_total || !launches) { set_error("v2p_env_profile_end: bad argument"); return V2P_ERR_INVALID; }
    if (!e->prof_ev) { set_error("v2p_env_profile_end: no measurement is open"); return V2P_ERR_INVALID; }
    DeviceGuard g(e->device);
    double total = 0.0;
    int rc = V2P_OK;
    for (int64_t k = 0; k < e->prof_n && rc == V2P_OK; ++k) {
        float ms = 0.f;
        rc = check_hip(hipEventSynchronize(e->prof_ev[2 * k + 1]), "hipEventSynchronize");
        if (rc == V2P_OK) rc = check_hip(hipEventElapsedTime(&ms, e->prof_ev[2 * k], e->prof_ev[2 * k + 1]), "hipEventElapsedTime");
        total += ms;
    }
    *physics_ms_total = total;
    *launches = e->prof_n;
    profile_free(e);
    return rc;
}

int v2p_env_export(v2p_env* e, void* stream) {
    if (!e) { set_error("v2p_env_export: bad argument"); return V2P_ERR_INVALID; }
    if (e->schedule == 0) return V2P_OK;  // the link-per-lane kernel writes the exposed tensors itself
    DeviceGuard g(e->device);
    return launch_env_export(e, (hipStream_t)stream);
}

int v2p_env_post_physics(v2p_env* e, void* stream) {
    if (!e) { set_error("v2p_env_post_physics: bad argument"); return V2P_ERR_INVALID; }
    DeviceGuard g(e->device);
    return launch_env_post(e, (hipStream_t)stream);
}

int v2p_env_step(v2p_env* e, float* actions, void* stream) {
    if (e && actions && e->schedule == 0) {
        // link-per-lane schedule: pre-physics runs in the physics kernel's prologue (lane = link owns its joint's action components)
        DeviceGuard g(e->device);
        // ... and post-physics in the epilogue of every env's last job, where the kernel instantiation has it (else its own kernel)
        int fused = 0;
        int rc = physics_launch(e, (hipStream_t)stream, actions, &fused);
        if (rc == V2P_OK && fused) e->cur_target = 1 - e->cur_target;
        else if (rc == V2P_OK) rc = launch_env_post(e, (hipStream_t)stream);
        return rc;
    }
    int rc = v2p_env_pre_physics(e, actions, stream);
    if (rc == V2P_OK) rc = v2p_env_physics(e, stream);
    if (rc == V2P_OK) rc = v2p_env_export(e, stream);
    if (rc == V2P_OK) rc = v2p_env_post_physics(e, stream);
    return rc;
}

int v2p_env_push_state(v2p_env* e, const int64_t* env_ids, int64_t n, int with_rb_state, void* stream) {
    if (!e || n < 0 || n > e->n) { set_error("v2p_env_push_state: bad argument"); return V2P_ERR_INVALID; }
    DeviceGuard g(e->device);
    return launch_env_push_state(e, env_ids, env_ids ? n : e->n, with_rb_state, (hipStream_t)stream);
}

int v2p_env_set_schedule(v2p_env* e, int schedule) {
    if (!e || (schedule != 0 && schedule != 1)) { set_error("v2p_env_set_schedule: bad argument"); return V2P_ERR_INVALID; }
    if (schedule == 1 && e->num_shapes > 1) { set_error("v2p_env_set_schedule: the env-per-lane kernel handles single-shape batches only"); return V2P_ERR_UNSUPPORTED; }
    if (schedule == 1 && e->p.solver_type == 1) { set_error("v2p_env_set_schedule: the env-per-lane cross-check kernel solves PGS only"); return V2P_ERR_UNSUPPORTED; }
    if (schedule == 1 && e->p.friction_frame == 1) { set_error("v2p_env_set_schedule: the env-per-lane cross-check kernel solves in the world friction frame only"); return V2P_ERR_UNSUPPORTED; }
    if (schedule == 1 && e->p.joint_limits) { set_error("v2p_env_set_schedule: the env-per-lane cross-check kernel has no joint limits"); return V2P_ERR_UNSUPPORTED; }
    if (schedule == 1) {
        DeviceGuard g(e->device);
        int rc = ensure_env_per_lane_buffers(e);
        if (rc != V2P_OK) return rc;
    }
    e->schedule = schedule;
    return V2P_OK;
}

int v2p_env_target_index(const v2p_env* e) { return e ? e->cur_target : V2P_ERR_INVALID; }
int v2p_env_kernel_build(const v2p_env* e) {  // (the build the NEXT launch of the batch runs)
    if (!e) return V2P_ERR_INVALID;
    if (e->kernel_build == 0 && !e->build_latched) return build_wanted(e) ? 2 : 1;  // (read-only: what choose_build would take now)
    return e->ll_regs_build ? 2 : 1;
}

int v2p_env_debug_contacts(v2p_env* e, int32_t* out, void* stream) {
    if (!e || !out) { set_error("v2p_env_debug_contacts: bad argument"); return V2P_ERR_INVALID; }
    if (!e->contact_ids) { set_error("v2p_env_debug_contacts: the batch was created with v2p_sim_cfg.debug_contacts = 0"); return V2P_ERR_INVALID; }
    DeviceGuard g(e->device);
    return check_hip(hipMemcpyAsync(out, e->contact_ids, sizeof(int32_t) * NB * 4 * (size_t)e->n, hipMemcpyDeviceToDevice, (hipStream_t)stream),
                     "hipMemcpyAsync(contact_ids)");
}

int v2p_env_debug_contacts_substeps(v2p_env* e, int32_t* out, void* stream) {
    if (!e || !out) { set_error("v2p_env_debug_contacts_substeps: bad argument"); return V2P_ERR_INVALID; }
    if (!e->contact_ids_sub) { set_error("v2p_env_debug_contacts_substeps: the batch was created with v2p_sim_cfg.debug_contacts < 2"); return V2P_ERR_INVALID; }
    DeviceGuard g(e->device);
    return check_hip(hipMemcpyAsync(out, e->contact_ids_sub, sizeof(int32_t) * NB * 4 * (size_t)e->n * (size_t)e->p.nsub, hipMemcpyDeviceToDevice, (hipStream_t)stream),
                     "hipMemcpyAsync(contact_ids_sub)");
}

int v2p_env_debug_pairing(v2p_env* e, int32_t* perm, int32_t* key, void* stream) {
    if (!e || !perm || !key) { set_error("v2p_env_debug_pairing: bad argument"); return V2P_ERR_INVALID; }
    DeviceGuard g(e->device);
    int rc = V2P_OK;
    if (env_pairing_on(e) && e->pair_have) rc = launch_env_pairing(e, (hipStream_t)stream);
    if (rc == V2P_OK) rc = check_hip(hipMemcpyAsync(perm, e->perm, sizeof(int32_t) * (size_t)e->n, hipMemcpyDeviceToDevice, (hipStream_t)stream), "hipMemcpyAsync(perm)");
    if (rc == V2P_OK) rc = check_hip(hipMemcpyAsync(key, e->pair_key, sizeof(int32_t) * (size_t)e->n, hipMemcpyDeviceToDevice, (hipStream_t)stream), "hipMemcpyAsync(pair_key)");
    return rc;
}

}  // extern "C"
