// Internal (non-ABI) declarations shared by the translation units of libv2p_rollout.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/v2p_rollout.h"

namespace v2p {

constexpr int NB = V2P_NUM_BODIES;
constexpr int NJ = NB - 1;
constexpr int NDOF = V2P_NUM_DOF;
constexpr int NACT = V2P_NUM_ACTIONS;
constexpr int NOBS = V2P_NUM_OBS;
constexpr int MSD = V2P_MOTION_STATE_DIM;
constexpr int MAX_HULL_VERTS = 1536;
constexpr int HULL_PAD = 8;
constexpr int MAX_DEPTH = 12;
constexpr int MAX_BRANCH = 3;  // links with more than one child (root + chest for SMPL)

// offsets inside one packed motion-state row [331]
constexpr int MS_ROOT_POS = 0, MS_ROOT_ROT = 3, MS_DOF_POS = 7, MS_ROOT_VEL = 76, MS_ROOT_ANG_VEL = 79, MS_DOF_VEL = 82, MS_KEY_POS = 151,
              MS_RB_POS = 163, MS_RB_ROT = 235;

// Numeric data of one body SHAPE (the reference builds one asset per clip from its SMPL betas, humanoid_smpl_im.py:279-296):
// every env points at one of these; the tree itself is the same for all shapes.
struct DevShape {
    float local_pos[NB][3];
    float mass[NB];
    float com[NB][3];
    float inertia[NB][6];  // xx xy xz yy yz zz about COM, body axes
    float kp[NB];          // per joint (isotropic over its 3 axes), index = body id, [0] unused
    float kd[NB];
    float arm[NB];
    float bound_radius[NB];  // max |hull vertex| (contact culling, env-per-lane kernel)
    float aabb_c[NB][3], aabb_e[NB][3];  // body-frame bounding box of the hull: centre, half extents (tighter culling)
    int32_t hull_offsets[NB + 1];  // into hull_verts; every body's list is padded to a multiple of HULL_PAD (last vertex repeated)
    int32_t hull_count[NB];        // real vertex count per body
    int32_t hull_cofs[NB + 1];     // offsets of the unpadded lists (LDS copy of the link-per-lane kernel)
    float hull_verts[MAX_HULL_VERTS][3];
    float limit_lo[NB][3], limit_hi[NB][3];  // per-DOF range of the joint's exponential-map coordinate (radians), index = body id
};

// Body model as the kernels see it: the tree (uniform across lanes, scalar loads) + the shape of a single-shape batch.
struct DevModel {
    int32_t parents[NB];
    int32_t depth[NB];
    int32_t order[NB];     // links in level (breadth-first) order: consecutive entries never depend on each other's results
    int32_t children[NB][3];  // up to 3 children per link, -1 = none (lane = link kernel)
    int32_t anc_mask[NB];     // bit i set when link i is an ancestor of (or is) the link
    int32_t desc_mask[NB];    // bit i set when link i is in the subtree of the link (self included)
    int32_t max_depth;
    int32_t multi_child_levels;  // bit d set when some link at depth d-1 has more than one child
    int32_t max_hull_count;
    int32_t nonchain_levels;     // bit d set when some link at depth d does not directly follow its parent (parent != link - 1)
    int32_t lam_slot[NB];  // index into the saved-Lambda register sets for branching links (root = 0), -1 otherwise
    int32_t anc_jump[NB];     // ancestors 1, 2, 4, 8 levels up, 8 bits each (255 = none): kinematics by pointer doubling
    int32_t jump_rounds;      // ceil(log2(max_depth + 1)), at most 4
    int32_t side_depths[NB];  // bit d set when the ancestor (or self) of the link at depth d is not the FIRST child of its parent
    DevShape shape;
};

// the model is immutable while kernels run: reading it through the constant address space keeps every
// access a scalar load that the compiler may hoist and batch (stores to the workspace cannot clobber it)
typedef const DevModel __attribute__((address_space(4))) ConstModel;
typedef const DevShape __attribute__((address_space(4))) ConstShape;

struct DevTables {
    v2p_motion_tables t;
};

}  // namespace v2p

struct v2p_model {
    v2p::DevModel host;
    v2p::DevModel* dev;
    int device;
};

struct v2p_mlib {
    v2p_motion_tables t;
    int device;
};

namespace v2p {

// physics workspace layout: see physics.hip
struct EnvParams {
    float h;               // substep length
    int nsub;              // substeps per control step
    int hold_sub;          // substeps during which the residual wrench acts
    int n_iter;
    int enable_contact;
    float gravity_z, mu, contact_offset, max_depen, erp, ang_damp, max_ang_vel;
    float pd_tar_lim, res_force_scale, res_torque_scale, ground_tolerance, max_episode_length;
    int enable_early_termination;
    int freeze_terminated;  // envs whose reset flag is set are not simulated (their state stays as it is)
    int solver_type;        // 0 PGS, 1 TGS (frozen Jacobians)
    int joint_limits;       // 1: limit rows for DOFs with a range narrower than a full turn
    float limit_margin;     // ... that exist only while C < limit_margin + h max(0, approach rate of v*)
    float rest_offset;      // gap of a hull-vertex row = z - rest_offset (sim.physx.rest_offset)
    float bounce_threshold; // sim.physx.bounce_threshold_velocity (the humanoid's rows have restitution 0: kept for the record)
    int friction_frame;     // 0 world (t1 = x, t2 = y), 1 velocity (t1 along the tangential velocity of the point under v*): hull x ground rows
    int context_length, context_padding;
    float dt;              // control step
    float term_heights[NB];
    float body_pos_weights[NB];
    float reward_specs[8];
    float aug[NB];         // joint-diagonal augmentation armature + h kd + h^2 kp per link
};

}  // namespace v2p

namespace v2p { struct BallDev; }

struct v2p_env {
    const v2p_model* model;   // shape 0 (the tree tables of every shape are identical)
    int num_shapes;           // > 1: per-env body shapes
    v2p::DevShape* shapes_dev;   // [num_shapes] (multi-shape only)
    int32_t* env_shape_dev;      // [N]
    float* shape_aug_dev;        // [num_shapes][24]
    const v2p_mlib* mlib;
    v2p::EnvParams p;
    v2p_env_buffers buf;
    int64_t n;
    int device;
    int cur_target;           // index of the current target buffer
    int schedule;             // 0 = link per lane (physics_ll.hip), 1 = env per lane (physics.hip)
    const int64_t* motion_id; // [N] device (borrowed)
    float* state;             // [N][STATE_SLOTS]
    float* ctrl;              // [N][CTRL_SLOTS]: pd target 69, wrench 6
    float* out;               // [N][OUT_SLOTS]: physics outputs before export
    float* ws;                // SoA [WS_SLOTS][N] physics workspace
    int32_t* contact_ids;     // [N,24,4] debug
    int32_t* contact_ids_sub; // [N,nsub,24,4] debug, every substep (v2p_sim_cfg.debug_substep_contacts), else NULL
    long long* prof;          // [8] phase cycle counters when V2P_PHASE_TIMING is set (device), else NULL
    long long* wave_times;    // [waves][4] per-wave wall-clock stamps of the last launch when V2P_WAVE_TIMES=<file> is set
    // pairing (physics_ll.hip): envs are handed to waves in descending order of their contact load
    int32_t* pair_key;        // [N] load key of each env after the last physics launch (0..255)
    int32_t* pair_pos;        // [N] arrival index inside its load bin
    int32_t* pair_hist;       // [256] + pair_start [256] + pair_done [1] (one allocation)
    int32_t* pair_start;
    int32_t* pair_done;
    int32_t* perm;            // [N] wave slot -> env of the next physics launch, materialised for v2p_env_debug_pairing only
    int32_t* pair_list[2];    // [256][N] envs of each load bin in arrival order: what the NEXT launch looks its envs up in (double
    int32_t* pair_starts[2];  // [256]    first rank of each bin                  buffered: a launch reads one set and fills the other)
    int pair_buf;             // the set the next launch reads
    int job_mono_default;     // job_mono_permille was left at its default (v2p_env_attach_ball moves it)
    int32_t* pair_slot_env;   // [N] env of each wave slot of the running launch: looked up by the job of the first substep, read by the later ones
    int pair_period;          // 0 = pairing off (v2p_sim_cfg.pair_envs_by_load = 0), else on
    int substeps_per_sim;     // substeps of one simulate() call
    int substep_jobs;         // v2p_sim_cfg.substep_jobs: the physics launch is cut into (substep, env pair) jobs
    int job_min_blocks;       // ... when it has more env pairs than this (0: always)
    int32_t* job_progress;    // [waves + 1] progress word per wave slot, last = error flag
    float* job_hand;          // [nsub - 1][N][HAND_FLOATS] the state as one substep job hands it to the next (16-byte chunks), a slot per substep
    long job_timeout_spins;   // see PhysArgs
    int job_interleave;
    int job_len;              // substeps per job; 0 = the engine decides (2 for launches of >= job_len2_blocks env pairs, else 1)
    int job_len2_blocks;
    int ll_regs_build;        // 1: this batch runs the register build of the link-per-lane kernel (two waves per SIMD)
    int kernel_build;         // v2p_sim_cfg.kernel_build (0: ll_regs_build follows the envs resident on the device)
    int build_latched;        // kernel_build 0: the choice is taken at the first launch after creation / after a whole-batch reset and holds until the next one
    int counted_resident;     // this batch is in the device's resident-env count
    int job_lead;             // substeps of the FIRST job of a cut pair (0 = like the others, -1 = the engine decides)
    int64_t job_recoveries;   // jobs that gave up waiting and recomputed, as last fetched (v2p_env_check / _check_async)
    int64_t jobs_skipped;     // late jobs that found their pair's step complete and did not run (their per-call records are missing for that step)
    int64_t jobs_skipped_reported;
    int job_epoch;
    int pair_mix_permille;    // share of the envs (the heaviest) that are paired with the lightest ones instead of with each other
    int pair_mix_default;     // pair_mix_permille was left to the engine (-1)
    int job_mono_permille;    // share of the env pairs (the heaviest) whose substeps stay in one workgroup
    v2p::BallDev* ball;       // racket + ball attached (v2p_env_attach_ball), else NULL
    int32_t* err_host;        // pinned copy of the substep jobs' error word (v2p_env_check_async), lazily allocated
    hipEvent_t err_event;
    int err_pending;
    hipEvent_t* prof_ev;      // 2 events per measured physics launch (v2p_env_profile_begin), else NULL
    int64_t prof_cap, prof_n;
    int64_t prof_seen;        // physics launches since v2p_env_profile_begin
    int32_t prof_stride, prof_period;  // which of them are bracketed (v2p_env_profile_begin_sampled)
    int pair_have;            // the last physics launch left (key, pos, start) that have not been scattered into perm yet
};

#ifndef V2P_LL_WPB
#define V2P_LL_WPB 1   // waves per workgroup of physics_ll_kernel
#endif

namespace v2p {

// wave slots (env pairs) of a physics_ll launch = progress words of the substep jobs; the error word sits right behind them
inline int64_t job_wave_slots(int64_t n) { return (n + 2 * V2P_LL_WPB - 1) / (2 * V2P_LL_WPB) * V2P_LL_WPB; }

// state SoA slots
constexpr int ST_ROOT_POS = 0, ST_ROOT_QUAT = 3, ST_JQUAT = 7, ST_VEL = 7 + 4 * NJ, STATE_SLOTS = 7 + 4 * NJ + 6 + 3 * NJ;  // 174
constexpr int HAND_FLOATS = 224;  // 56 chunks of 4 floats: chunk 2b, 2b+1 = joint b (quaternion | rate), 0, 1, 48, 49 = root (49 also: residual force), 50 .. 53 = the ball (state 13 | aerodynamic force 3), 54 = residual torque
constexpr int CT_PD = 0, CT_FORCE = NDOF, CT_TORQUE = NDOF + 3, CTRL_SLOTS = NDOF + 6;
// physics outputs (SoA): rigid-body state 24x13, dof_pos 69, contact force 72, dof force 69
constexpr int OUT_RB = 0, OUT_DOF_POS = NB * 13, OUT_CONTACT = OUT_DOF_POS + NDOF, OUT_DOF_FORCE = OUT_CONTACT + NB * 3,
              OUT_SLOTS = OUT_DOF_FORCE + NDOF;

// engine-owned arrays are ENV-MAJOR: [env][slot].  One workgroup of the link-per-lane kernel owns whole envs, so its
// loads/stores stay inside one XCD's L2 and merge into full lines (slot-major put every line under 8 XCDs: 4x write traffic)
#define SIDX(slot) ((int64_t)e * v2p::STATE_SLOTS + (slot))
#define CIDX(slot) ((int64_t)e * v2p::CTRL_SLOTS + (slot))
#define OIDX(slot) ((int64_t)e * v2p::OUT_SLOTS + (slot))

void set_error(const char* fmt, ...);
const char* debug_env(const char* name);  // getenv in a process that sets V2P_DEBUG=1, else NULL (profiling switches only)
int check_hip(hipError_t e, const char* what);

// launchers (each in its own translation unit)
int launch_motion_state(const v2p_motion_tables& t, const int64_t* ids, const float* times, int64_t q, int adjust_height, float ground_tol,
                        float* const out[9], hipStream_t s);
int launch_reward(int64_t n, const float* body_pos, const float* body_rot, const float* tgt_pos, const float* tgt_rot, const float* dof_pos,
                  const float* dof_vel, const float* tgt_dof_pos, const float* tgt_dof_vel, const float* w, const float* specs, float* rew,
                  float* sub, hipStream_t s);
int launch_reset_flags(int64_t n, const int64_t* progress, const float* rb_pos, const float* heights, const float* cur_time,
                       const float* clip_len, float max_len, int early, int64_t* reset_out, int64_t* term_out, hipStream_t s);
int launch_obs_imitation(int64_t n, const float* body_pos, const float* body_rot, const float* tgt_pos, const float* tgt_rot,
                         const float* dof_pos, const float* dof_vel, const float* tgt_dof_pos, const float* body_vel,
                         const float* body_ang_vel, const float* motion_bodies, const float* nmean, const float* nstd, float nclip, float* obs,
                         hipStream_t s);
int launch_obs_imitation_packed(int64_t rows, int64_t steps, const float* obs461, const float* context_feat, int64_t ctx_frames, int64_t first_frame,
                                const float* nmean, const float* nstd, float nclip, float* obs, hipStream_t s);
int launch_policy_head(int64_t n, float* mu, const float* context_feat, int64_t ctx_frames, int64_t frame, const float* logstd, const float* noise,
                       float* action, float* sigma_out, float* neglogp, hipStream_t s, float* action_row = nullptr, float* mu_row = nullptr);
int launch_gae(int64_t horizon, int64_t n, const float* fdones, const float* values, const float* rewards, const float* next_values, float gamma,
               float tau, float* advs, hipStream_t s);
int launch_motion_tables_build(int64_t F, const double* lrot, const double* root_trans, const int32_t* frame_clip, const int64_t* clip_start,
                               const int32_t* clip_frames, const double* clip_dt, const int32_t* parents_host, const double* local_pos, int per_clip,
                               float* gts, float* grs, float* lrs, float* grvs, float* gravs, float* dvs, hipStream_t s);
int launch_shape_compile(int32_t jobs, const double* pts, const int32_t* job_off, int32_t max_pts, const double* dirs, const int32_t* dir_off, int32_t num_tables,
                         double density, int32_t max_verts, double eps_rel, double* mass, double* com, double* inertia, int32_t* num_verts, int32_t* vert_ids,
                         double* verts, int32_t* status, hipStream_t s);
int launch_value_record(int64_t n, const float* x, const double* mean, const double* var, float eps, const float* terminated, float* values_row,
                        float* next_values_row, hipStream_t s);
int launch_rollout_record(int64_t n, const float* obs, int64_t obs_dim, const float* rew, const int64_t* reset, const int64_t* terminate, const float* sub_rewards,
                          float* next_obs_row, float* rewards_row, float* dones_row, float* dones, float* terminated, float* prev_dones, float* cur_rewards,
                          float* cur_lengths, double* acc, double* sub_acc, hipStream_t s);
int launch_env_reset(v2p_env* e, const int64_t* env_ids, int64_t n, const float* motion_times, hipStream_t s);
int launch_env_context(v2p_env* e, const int64_t* env_ids, int64_t n, const float* motion_times, hipStream_t s);
int launch_env_pre(v2p_env* e, float* actions, hipStream_t s);
int launch_env_physics(v2p_env* e, hipStream_t s);
// actions: fuse pre-physics into the kernel; fused_post (with actions): non-null = post-physics may be fused in as well, *fused_post says whether it was
int launch_env_physics_ll(v2p_env* e, hipStream_t s, float* actions = nullptr, int* fused_post = nullptr);
bool env_pairing_on(const v2p_env* e);
struct PairView;
PairView env_pair_view(const v2p_env* e);
int launch_env_pairing(v2p_env* e, hipStream_t s);  // scatter (key, pos, start) -> perm when env_pre_kernel has not done it
int launch_env_export(v2p_env* e, hipStream_t s);
int ensure_env_per_lane_buffers(v2p_env* e);  // the env-per-lane schedule's global workspace, allocated on first use
int launch_env_post(v2p_env* e, hipStream_t s);
int launch_env_push_state(v2p_env* e, const int64_t* env_ids, int64_t n, int with_rb, hipStream_t s);
int physics_ws_slots();

}  // namespace v2p
