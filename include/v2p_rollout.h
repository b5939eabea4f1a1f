/*
 * v2p_rollout.h -- C ABI of the MI355X-native rollout engine for vid2player3d's
 * embodied_pose SMPL-humanoid imitation task.
 *
 * This is the drop-in boundary.  The reference's task object
 * (embodied_pose/env/tasks/humanoid_smpl_im.py) talks to Isaac Gym through ~25 gym.* calls;
 * the entry points below are what a reference-side binding would call instead.  Each entry
 * point names the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C, no torch types.  Every `float*`/`int64_t*` is a DEVICE pointer owned by the
 *     caller (PyTorch-ROCm tensors on the Python side); the library never frees them.
 *   - row-major, fp32 data, int64 flags/indices (reference buffer dtypes, base_task.py:62-74),
 *     quaternions xyzw.
 *   - every call returns 0 on success or a negative v2p_status; v2p_last_error() returns a
 *     thread-local message.  The Python shim turns non-zero codes into RuntimeError (the
 *     reference's error convention is Python exceptions).
 *   - calls are asynchronous on the hipStream_t passed as `void* stream` (NULL = default
 *     stream); a handle is single-threaded; handles on different GPUs are independent.
 *   - B = 24 bodies, D = 69 dofs, A = 75 actions, OBS = 461.
 */
#ifndef V2P_ROLLOUT_H
#define V2P_ROLLOUT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V2P_NUM_BODIES 24
#define V2P_NUM_DOF 69
#define V2P_NUM_ACTIONS 75
#define V2P_NUM_OBS 461
#define V2P_MOTION_STATE_DIM 331  /* root_pos3 root_rot4 dof_pos69 root_vel3 root_ang_vel3 dof_vel69 key_pos12 rb_pos72 rb_rot96 */
#define V2P_CONTEXT_DIM 378       /* body_pos72 body_rot96 dof_pos69 body_pos_gt72 dof_pos_gt69 (humanoid_smpl_im.py:202) */
#define V2P_ABI_VERSION 14

typedef enum {
    V2P_OK = 0,
    V2P_ERR_INVALID = -1,     /* bad argument */
    V2P_ERR_UNSUPPORTED = -2, /* valid in the reference, not built yet (e.g. anisotropic joint gains) */
    V2P_ERR_HIP = -3,         /* HIP runtime error */
    V2P_ERR_NOMEM = -4,
    V2P_ERR_INTERNAL = -5     /* the engine noticed that a launch did not do all it should have (v2p_env_check: skipped substep jobs) */
} v2p_status;

typedef struct v2p_model v2p_model; /* body model (host copy + device constants) */
typedef struct v2p_mlib v2p_mlib;   /* reference-motion tables (device pointers, not owned) */
typedef struct v2p_env v2p_env;     /* a batch of environments on one GPU */

/* ---- body model -------------------------------------------------------------------------
 * Replaces gym.load_asset / create_actor / get+set_actor_dof_properties /
 * get_actor_rigid_body_properties (humanoid_smpl_im.py:273-287, 356-389): the caller hands
 * over the compiled model (HOST pointers; copied). */
typedef struct {
    int32_t num_bodies;          /* must be 24 */
    const int32_t* parents;      /* [B]   -1 for the root */
    const float* local_pos;      /* [B,3] joint offset in the parent frame */
    const float* mass;           /* [B] */
    const float* com;            /* [B,3] body frame */
    const float* inertia;        /* [B,9] about COM, body axes */
    const float* kp;             /* [D] already scaled by body mass (humanoid_smpl_im.py:376-385) */
    const float* kd;             /* [D] */
    const float* armature;       /* [D] */
    const int32_t* hull_offsets; /* [B+1] */
    const float* hull_verts;     /* [V,3] body frame */
    /* ---- ABI 7 */
    const float* limit_lower;    /* [D] nullable: per-DOF range of the exponential-map coordinate, radians (the MJCF `range` that */
    const float* limit_upper;    /* [D] nullable   gym.get_actor_dof_properties reports, humanoid_smpl.py:318-337); NULL = unlimited */
} v2p_model_desc;

int v2p_model_create(const v2p_model_desc* desc, int device, v2p_model** out);
void v2p_model_destroy(v2p_model* m);

/* ---- reference-motion tables ------------------------------------------------------------
 * Replaces the tensor attributes of utils/motion_lib.py:MotionLib (motion_lib.py:78-99,
 * 370-384).  DEVICE pointers, borrowed for the lifetime of the handle. */
typedef struct {
    int64_t num_motions;
    int64_t num_frames_total;
    const float* gts;               /* [F,24,3] global translations */
    const float* grs;               /* [F,24,4] global rotations */
    const float* lrs;               /* [F,24,4] local rotations */
    const float* grvs;              /* [F,3]  root linear velocity */
    const float* gravs;             /* [F,3]  root angular velocity */
    const float* dvs;               /* [F,69] dof velocities */
    const float* motion_lengths;    /* [C] seconds */
    const int64_t* motion_num_frames; /* [C] */
    const float* motion_dt;         /* [C] */
    const float* motion_min_verts_h; /* [C] */
    const int64_t* length_starts;   /* [C] first row of each clip */
    const float* motion_bodies;     /* [C,11] gender + 10 betas */
    int32_t key_body_ids[4];        /* amass_im.yaml:17 -> R_Ankle, L_Ankle, L_Hand, R_Hand */
} v2p_motion_tables;

int v2p_mlib_create(const v2p_motion_tables* tables, int device, v2p_mlib** out);
void v2p_mlib_destroy(v2p_mlib* m);

/* MotionLib.get_motion_state(..., return_rigid_body=True) (motion_lib.py:164-266).
 * `out` points to 9 device arrays in the order root_pos[Q,3] root_rot[Q,4] dof_pos[Q,69]
 * root_vel[Q,3] root_ang_vel[Q,3] dof_vel[Q,69] key_pos[Q,4,3] rb_pos[Q,24,3] rb_rot[Q,24,4];
 * NULL entries are skipped. */
int v2p_motion_state(const v2p_mlib* m, const int64_t* motion_ids, const float* motion_times, int64_t num_queries,
                     int adjust_height, float ground_tolerance, float* const out[9], void* stream);

/* ---- stand-alone task ops (op-level parity tests; also usable by an unmodified task) -----
 * compute_humanoid_reward (humanoid_smpl_im.py:918-953); specs = k_dof,k_vel,k_pos,k_rot,
 * w_dof,w_vel,w_pos,w_rot. */
int v2p_reward(int64_t n, const float* body_pos, const float* body_rot, const float* tgt_pos, const float* tgt_rot,
               const float* dof_pos, const float* dof_vel, const float* tgt_dof_pos, const float* tgt_dof_vel,
               const float* body_pos_weights /*[24]*/, const float specs[8], float* reward /*[n]*/, float* sub_rewards /*[n,4]*/,
               void* stream);
/* compute_humanoid_reset (humanoid_smpl_im.py:956-987); contact bodies are excluded through
 * `term_heights_masked` = termination heights with -inf at the contact bodies. */
int v2p_reset_flags(int64_t n, const int64_t* progress, const float* rb_pos, const float* term_heights_masked /*[24]*/,
                    const float* cur_time, const float* clip_len, float max_episode_length, int enable_early_termination,
                    int64_t* reset_out, int64_t* terminated_out, void* stream);
/* compute_humanoid_observations_imitation, the 734-d in-network observation
 * (humanoid_smpl_im.py:773-850 == models/im_network_builder.py:262-338), local_root_obs =
 * root_height_obs = True. */
int v2p_obs_imitation(int64_t n, const float* body_pos, const float* body_rot, const float* tgt_pos, const float* tgt_rot,
                      const float* dof_pos, const float* dof_vel, const float* tgt_dof_pos, const float* body_vel,
                      const float* body_ang_vel, const float* motion_bodies /*[n,11]*/,
                      const float* norm_mean /*[734] nullable*/, const float* norm_std /*[734] nullable*/, float norm_clip,
                      float* obs /*[n,734]*/, void* stream);
/* ... with RunningNorm.forward in eval mode fused when norm_mean/norm_std are given: clamp((x-mean)/(std+1e-8), +-clip)
 * (models/running_norm.py:32-43). */

/* The same features straight from what the policy network is handed (models/im_network_builder.py:150-189, preprocess_input +
 * compute_humanoid_obs): `obs` rows [rows,461] as the task packs them (humanoid_smpl_im.py:198) and the context frames
 * [envs,ctx_frames,378] (humanoid_smpl_im.py:202, body_pos | body_rot | dof_pos | ...).  rows = envs * steps; row env*steps + k is
 * paired with context frame first_frame + k: rollout (eval) = steps 1, first_frame = context_padding + t; training minibatches
 * (flatten=True) = steps T, first_frame = context_padding.  No split / view / cat of the inputs is materialised. */
int v2p_obs_imitation_packed(int64_t rows, int64_t steps, const float* obs /*[rows,461]*/, const float* context_feat /*[rows/steps,ctx_frames,378]*/,
                             int64_t ctx_frames, int64_t first_frame, const float* norm_mean /*nullable*/, const float* norm_std /*nullable*/,
                             float norm_clip, float* out /*[rows,734]*/, void* stream);

/* Policy head of the rollout (ABI 9): what the reference does between the actor MLP and the env step.
 *   models/im_network_builder.py:219-228 (eval_actor, residual_action = True): mu[:, :69] += cur_context['dof_pos'] - the target DOF
 *     positions of context frame `frame` (= context_padding + t in the rollout), read straight from context_feat [n,ctx_frames,378];
 *   models/im_models.py:45-48: action = Normal(mu, sigma).sample() = mu + sigma * noise, neglogp (rl_games ModelA2CContinuousLogStd):
 *     0.5 sum(((a - mu) / sigma)^2) + 0.5 log(2 pi) * 75 + sum(logstd), sigma = exp(logstd) (fixed_sigma, cfg/amass_im.yaml:76-81).
 * mu [n,75]: the actor MLP's output on entry, the residual mean on exit; noise [n,75] standard normal (the caller's generator);
 * action [n,75], sigma [n,75] (nullable), neglogp [n] are written. */
int v2p_policy_head(int64_t n, float* mu, const float* context_feat, int64_t ctx_frames, int64_t frame, const float* logstd /*[75]*/,
                    const float* noise, float* action, float* sigma, float* neglogp, void* stream);

/* v2p_policy_head with the experience buffer's rows as additional destinations (the rollout: ExperienceBuffer.update_data of actions / mus /
 * sigmas / neglogpacs, im_agent.py:348-352, without the four copy kernels): sigma_row [n,75] and neglogp_row [n] are written in place of
 * `sigma` / `neglogp`, action_row / mu_row [n,75] (nullable) receive copies of `action` / the residual mean.  `action` stays a tensor of
 * its own - env.step masks the rows of finished envs in place, the buffer keeps the sampled actions. */
int v2p_policy_head_record(int64_t n, float* mu, const float* context_feat, int64_t ctx_frames, int64_t frame, const float* logstd,
                           const float* noise, float* action, float* sigma_row, float* neglogp_row, float* action_row, float* mu_row, void* stream);

/* The critic's output of a rollout step into the experience buffer (im_agent.py:292-303, 355, 398): value_raw [n] (the network's output) is
 * un-normalised with the value normaliser (running_mean / running_var: DEVICE float64 scalars of rl_games' RunningMeanStd; both NULL = no
 * normaliser) - sqrt(var + epsilon) * clamp(x, -5, 5) + mean - and written to values_row [n] (nullable) and, times (1 - terminated [n]), to
 * next_values_row [n] (nullable): one launch instead of ~9 elementwise kernels and two copies. */
int v2p_value_record(int64_t n, const float* value_raw, const double* running_mean, const double* running_var, float epsilon,
                     const float* terminated, float* values_row, float* next_values_row, void* stream);

/* Bookkeeping of one rollout step after env.step (ImitatorAgent.play_steps, agents/im_agent.py:380-409) in one launch: rewards / dones /
 * next_obses rows of the experience buffer, dones / terminate as floats, running episode returns and lengths, and the episode statistics
 * the reference collects through .nonzero() on the host - here float64 device accumulators:
 *   acc[0] += #episodes finished at this step, acc[1] += their returns, acc[2] += their lengths, acc[3] += #envs not done before the
 *   step, acc[4] += their rewards; sub_acc[k] += their sub-rewards k.  prev_dones / cur_rewards / cur_lengths [n] are updated in place.
 * obs [n,obs_dim] -> next_obs_row (nullable: no copy); rew [n], reset / terminate [n] int64, sub_rewards [n,4]; all DEVICE pointers. */
int v2p_rollout_record(int64_t n, const float* obs, int64_t obs_dim, const float* rew, const int64_t* reset, const int64_t* terminate,
                       const float* sub_rewards, float* next_obs_row, float* rewards_row, float* dones_row, float* dones, float* terminated,
                       float* prev_dones, float* cur_rewards, float* cur_lengths, double* acc /*[>=5]*/, double* sub_acc /*[4]*/, void* stream);

/* GAE reverse scan of the PPO rollout, CommonAgent.discount_values (learning/common_agent.py:423-435):
 * fdones [T,N], values / rewards / next_values / advs [T,N,1] (device). */
int v2p_gae(int64_t horizon, int64_t n, const float* fdones, const float* values, const float* rewards, const float* next_values,
            float gamma, float tau, float* advs, void* stream);

/* ---- reference-motion tables from clips ----------------------------------------------------
 * What the reference computes offline per clip with poselib (uhc/utils/convert_amass_isaac.py:134-153: SkeletonState.from_rotation_and_
 * root_translation -> SkeletonMotion.from_skeleton_state, poselib/poselib/skeleton/skeleton3d.py:409-431 forward kinematics, :1226-1249
 * finite-difference + Gaussian velocities) and MotionLib flattens (embodied_pose/utils/motion_lib.py:370-384, 443-458 dof velocities),
 * for all frames of all clips in two launches.  DEVICE pointers: local_rot [F,24,4] xyzw and root_trans [F,3] (float64, the clips
 * concatenated), frame_clip [F], clip_start [C] (= length_starts), clip_frames [C], clip_dt [C], local_pos [24,3] - or [C,24,3] with
 * per_clip_skeleton (every clip of the reference carries its own SMPL shape).  parents [24]: HOST array.  Out (float32, the layout of
 * v2p_motion_tables): gts [F,24,3] grs [F,24,4] lrs [F,24,4] grvs [F,3] gravs [F,3] dvs [F,69]. */
int v2p_motion_tables_build(int64_t num_frames_total, int64_t num_clips, const double* local_rot, const double* root_trans,
                            const int32_t* frame_clip, const int64_t* clip_start, const int32_t* clip_frames, const double* clip_dt,
                            const int32_t* parents, const double* local_pos, int32_t per_clip_skeleton, float* gts, float* grs, float* lrs,
                            float* grvs, float* gravs, float* dvs, void* stream);

/* ---- per-clip body assets ---------------------------------------------------------------
 * The geometry half of the reference's per-clip asset build (humanoid_smpl_im.py:255-296 -> uhc/smpllib/smpl_local_robot.py:79-143
 * `get_joint_geometries`: ConvexHull of every body's vertex cloud -> decimated mesh; Isaac Gym then integrates the mass properties at
 * the geom density), for thousands of (shape, body) JOBS in one launch, one wavefront per job, float64:
 *   cloud -> convex hull -> at most max_verts (<= 64) support vertices (the Fibonacci-direction tables `dirs`, tried in order until
 *   the distinct support points fit) -> hull of those -> mass, centre of mass, inertia about it.
 * All pointers are DEVICE pointers.  points [total,3]; job_offsets [num_jobs+1] (first point of every job); max_points = the
 * largest cloud (sizes the LDS arrays: ~48 B per point x 2; up to ~1600 points); dirs [.,3] + dir_offsets [num_dir_tables+1].
 * Out per job: mass, com [3], inertia [9], num_verts, vert_ids [max_verts] (indices into the job's cloud, ascending), verts
 * [max_verts,3], status (0 ok, 1 fewer than 4 points, 2 coplanar cloud, 3 face capacity, 4 reduction failed,
 * 7 the job's point count - job_off[j + 1] - job_off[j] - is negative or exceeds max_points).
 * vid2player3d_amd/body_shapes.py holds the host-side wrapper and the numpy statement of the same algorithm (the checker). */
int v2p_shapes_compile(int32_t num_jobs, const double* points, const int32_t* job_offsets, int32_t max_points, const double* dirs,
                       const int32_t* dir_offsets, int32_t num_dir_tables, double density, int32_t max_verts, double eps_rel, double* mass,
                       double* com, double* inertia, int32_t* num_verts, int32_t* vert_ids, double* verts, int32_t* status, void* stream);

/* ---- environments -----------------------------------------------------------------------
 * Simulation + task parameters (cfg/amass_im.yaml:3-52, utils/config.py:190-222). */
typedef struct {
    float sim_dt;               /* 1/60 (config.py:20) */
    int32_t substeps;           /* 2   (amass_im.yaml:38) */
    int32_t control_freq_inv;   /* 2   (amass_im.yaml:11)  -> control dt = 1/30, 4 physics substeps of 1/120 */
    int32_t num_solver_iterations; /* 4 (num_position_iterations) */
    int32_t enable_contact;     /* 0 = BASELINE config 2 (PD only) */
    int32_t residual_hold_sims; /* simulate() calls during which the residual wrench acts: 1 (first_sim) .. control_freq_inv (all) */
    float gravity_z;            /* -9.81 */
    float friction;             /* 1.0 */
    float contact_offset;       /* 0.02 */
    float max_depenetration_velocity; /* 10 */
    float erp;                  /* 0.2 */
    float angular_damping;      /* 0.01 (humanoid_smpl_im.py:274) */
    float max_angular_velocity; /* 100  (humanoid_smpl_im.py:275) */
    float pd_tar_lim;           /* 0.5*pi (humanoid_smpl_im.py:73) */
    float residual_force_scale; /* 31.85 */
    float residual_torque_scale;
    float ground_tolerance;     /* 0 */
    float max_episode_length;   /* 300 */
    int32_t enable_early_termination;
    int32_t context_length;     /* 32 */
    int32_t context_padding;    /* 8 */
    float term_heights[24];     /* per body; contact bodies get -inf (humanoid_smpl_im.py:217-224, 956-987) */
    float body_pos_weights[24]; /* humanoid_smpl_im.py:109-115 */
    float reward_specs[8];      /* k_dof,k_vel,k_pos,k_rot,w_dof,w_vel,w_pos,w_rot (humanoid_smpl_im.py:682) */
    int32_t freeze_terminated_envs; /* 0 (reference behaviour): envs whose reset flag is set keep being simulated as ragdolls until the
                                     * epoch reset, although nothing downstream reads them (zero reward, masked by `dones` in
                                     * im_agent.py:392-414).  1: their physics state is frozen instead (link-per-lane schedule). */
    /* ---- ABI 3 */
    int32_t schedule;           /* kernel schedule a batch starts with (v2p_env_set_schedule changes it): 0 = one link per lane (default),
                                 * 1 = one env per lane (cross-check) */
    int32_t pair_envs_by_load;  /* 1 (default): envs are handed to waves in descending order of their contact load; 0: in index order.
                                 * Results do not depend on it (tests), only the launch duration does. */
    int32_t solver_type;        /* contact solver: 0 = projected Gauss-Seidel (default), 1 = temporal Gauss-Seidel with frozen Jacobians
                                 * (sim.physx.solver_type of amass_im.yaml:41 is 1 = TGS in PhysX; see oracle/phys/v2p_phys_oracle.c for
                                 * what either means here).  Link-per-lane schedule only. */
    int32_t substep_jobs;       /* 1 (when the env pairs do not fit the GPU's wave slots in one round: > CUs x 8 pairs) or 2 (always): the
                                 * physics launch of the link-per-lane schedule is cut into (substep, env pair) jobs that hand the
                                 * state over through memory - 4x finer load balancing of the launch; results are bit-identical to 0
                                 * (one workgroup per env pair runs all substeps).  Every solver / contact setting of the link-per-lane schedule. */
    int32_t job_mono_permille;  /* substep_jobs: share of the env pairs (the heaviest) whose substeps stay in one workgroup; -1 = default (60; 250 above
                                 * 12288 envs, with joint limits or with a ball attached) */
    int32_t pair_mix_permille;  /* pair_envs_by_load: share of the envs (the heaviest) that share their wave with one of the lightest envs
                                 * instead of with an equally heavy one (a wave costs the union of its two envs' contact structure, and the
                                 * heaviest envs are the critical path of the launch); 0 = pairs of equals only; -1 = default (150, 500 with the register build - kernel_build; 0 above 12288 envs,
                                 * with joint limits or with a ball: measured) */
    int32_t debug_contacts;     /* diagnostics, off (0) by default: 1 = keep the contact vertex ids of the last substep
                                 * (v2p_env_debug_contacts; 384 B of extra stores per env-step), 2 = of EVERY substep as well
                                 * (v2p_env_debug_contacts_substeps) */
    /* ---- ABI 7 */
    int32_t joint_limits;       /* 1: every DOF whose range (v2p_model_desc.limit_lower/upper) is narrower than a full turn carries a
                                 * limit row in the contact solver (the racket arm of the player MJCFs; the amass MJCF has none).
                                 * Link-per-lane schedule, contacts on, either solver (under TGS the distance to the limit advances slice by
                                 * slice with the joint rate).  0 (default): ranges are ignored. */
    /* ---- ABI 9 */
    float limit_margin;         /* radians; <= 0 = default (0.05).  A limit row exists in a substep only while its DOF is within reach of the
                                 * limit: C < limit_margin + h * max(0, rate of approach after the unconstrained update v*), C = distance to
                                 * the nearer limit - the speculative activation every other row of the model has (contact_offset for hull
                                 * vertices, the closing distance of a substep for the ball), PhysX's `contactDistance` of a joint limit.
                                 * A joint far from its limits costs nothing; one that is pushed across by an impulse of the same substep
                                 * is caught in the next (C < 0 is always active) and corrected with erp.  1e9 = rows always on (ABI 8). */
    /* ---- ABI 11: the rest of the reference's sim.physx block (cfg/amass_im.yaml:39-48, utils/config.py:190-222) reaches the engine
     * instead of being parsed and dropped on the Python side */
    float rest_offset;          /* 0.0 (amass_im.yaml:45): distance at which a hull vertex rests on the plane - the gap of a hull-vertex row
                                 * is z - rest_offset (PhysX: contacts exist below contact_offset and come to rest at rest_offset) */
    float bounce_threshold_velocity; /* 0.2 (amass_im.yaml:46): approach speed below which a contact does not bounce.  The humanoid's
                                 * contacts have restitution 0 (plane.restitution 0.0), so nothing bounces at any speed; the ball rows take
                                 * their own copy (v2p_ball_cfg.bounce_threshold_velocity).  Must be >= 0. */
    int32_t num_velocity_iterations; /* 0 (amass_im.yaml:43).  The engine's solvers have no separate velocity pass: any other value is
                                 * REFUSED (V2P_ERR_UNSUPPORTED) rather than ignored. */
    /* ---- ABI 12 */
    int32_t kernel_build;       /* the library holds two builds of the link-per-lane physics kernel (same source, same results to float32
                                 * rounding): 1 = three waves per SIMD with the contact records parked in LDS (fastest where the launch is
                                 * bound by instruction issue: BASELINE's 8192 envs), 2 = two waves per SIMD with everything in registers
                                 * (5 - 7 % faster where a launch is as long as its heaviest env pair: small batches).  0 = the engine
                                 * chooses by the number of envs RESIDENT on the device - the sum over the live batches of this
                                 * process, so that rollout groups sharing a GPU are judged together (2 at <= 5120 envs: measured,
                                 * profiles/r04e_dual_build.txt).  The choice is taken at the first launch after v2p_env_create and
                                 * after every whole-batch v2p_env_reset (an epoch boundary) and holds in between: a batch does not
                                 * change build mid-epoch because another batch was created or destroyed.  v2p_env_kernel_build tells
                                 * which one the next launch of a batch runs (read-only). */
    /* ---- ABI 13: the A/B and test switches of the substep jobs (environment variables until ABI 12).  A zero-initialised block = the
     * engine's defaults.  The library reads NO engine option from the environment; its profiling switches (V2P_WAVE_TIMES,
     * V2P_PHASE_TIMING, V2P_PHASE_HEAVY, V2P_ENVS_PER_BLOCK) are honoured only in a process that sets V2P_DEBUG=1. */
    int32_t job_timeout_spins;  /* substep_jobs: polls (with back-off) a job waits for its predecessor before it recomputes the earlier
                                 * substeps itself; 0 = default (50000, ~20 ms), < 0 = give up at once (tests of the recovery path) */
    int32_t job_len;            /* substeps per job; 0 = the engine decides (1, or 2 for launches of >= CUs x 32 env pairs) */
    int32_t job_lead;           /* substeps of the FIRST job of a cut pair; 0 = the engine decides, < 0 = like the other jobs */
    int32_t job_no_interleave;  /* 1: the jobs of a pair are not interleaved with those of other pairs in dispatch order (A/B) */
    /* ---- ABI 14 */
    int32_t friction_frame;     /* tangent directions of the hull x ground rows.  0 (default) = world: t1 = x, t2 = y, each clamped to mu x the
                                 * normal impulse on its own - the friction limit is a box aligned with the world axes (a body pushed along the
                                 * diagonal holds up to sqrt(2) mu).  1 = velocity: t1 along the tangential velocity the contact point has under
                                 * the unconstrained velocity of the substep (where it would slide without contact impulses), t2 = n x t1; world
                                 * frame below 1e-6 m/s - the limit along the direction of sliding is mu whatever that direction is.  The
                                 * reference hands mu = 1 to PhysX (humanoid_smpl.py:175-182, amass_im.yaml:32-35), whose friction directions
                                 * follow the relative velocity at the contact; which of the two agrees with an Isaac Gym trace is for
                                 * tools/replay_trace.py --friction-frame to show.  Link-per-lane schedule (both solvers, ball, joint limits). */
} v2p_sim_cfg;

/* Caller-owned DEVICE buffers the engine reads/writes; these are the tensors the reference
 * task exposes (humanoid_smpl.py:66-113, base_task.py:62-74, humanoid_smpl_im.py:594-636). */
typedef struct {
    float* root_states;    /* [N,13] pos3 quat4 linvel3 angvel3 */
    float* dof_state;      /* [N,69,2] (pos, vel) interleaved like gym's dof state tensor */
    float* rb_state;       /* [N,24,13] */
    float* contact_force;  /* [N,24,3] net contact force per body */
    float* dof_force;      /* [N,69] */
    float* pd_target;      /* [N,69] clamped PD targets of the last step */
    float* obs;            /* [N,461] */
    float* rew;            /* [N] */
    float* sub_rewards;    /* [N,4] */
    int64_t* reset;        /* [N] */
    int64_t* terminate;    /* [N] */
    int64_t* progress;     /* [N] */
    float* cur_time;       /* [N] _cur_ref_motion_times */
    float* reset_time;     /* [N] _reset_ref_motion_times */
    float* target[2];      /* 2 x [N,331]: current / previous target motion state, flipped every step */
    float* context_feat;   /* [N,48,378] (nullable) */
    uint8_t* context_mask; /* [N,48]     (nullable) */
} v2p_env_buffers;

/* Replaces create_sim/add_ground/create_env/create_actor/prepare_sim/acquire_*_tensor
 * (humanoid_smpl.py:66-134, humanoid_smpl_im.py:231-389).  env_motion_id [N] is a DEVICE
 * array (each env is bound to one clip, humanoid_smpl_im.py:247-254). */
int v2p_env_create(const v2p_model* model, const v2p_mlib* mlib, const v2p_sim_cfg* cfg, const int64_t* env_motion_id,
                   int64_t num_envs, const v2p_env_buffers* buffers, int device, v2p_env** out);
/* The same with one body SHAPE per env: the reference builds one humanoid asset per sampled clip from its SMPL betas and
 * scale (humanoid_smpl_im.py:255-296, _create_smpl_humanoid_xml) and gives env i the asset of its clip.  `shapes` are
 * models with identical body trees, `env_shape_id` [N] is a HOST array of indices into `shapes` (copied). */
int v2p_env_create_shapes(const v2p_model* const* shapes, int32_t num_shapes, const int32_t* env_shape_id, const v2p_mlib* mlib,
                          const v2p_sim_cfg* cfg, const int64_t* env_motion_id, int64_t num_envs, const v2p_env_buffers* buffers,
                          int device, v2p_env** out);
void v2p_env_destroy(v2p_env* e);

/* HumanoidSMPL.reset(env_ids) with reference-state init (humanoid_smpl.py:136-173,
 * humanoid_smpl_im.py:442-563).  env_ids NULL = all envs.  motion_times [n] (device) are the
 * RSI phases; the caller draws them (MotionLib.sample_time, motion_lib.py:138-159) so that
 * the RNG stays on the Python side.  Fills state, target, obs, context. */
int v2p_env_reset(v2p_env* e, const int64_t* env_ids, int64_t n, const float* motion_times, void* stream);
/* HumanoidSMPLIM._init_context(motion_ids, motion_times) on its own (humanoid_smpl_im.py:530-563): the context window of the given envs
 * rebuilt around motion_times [n] (device) - frames motion_times + dt + dt * (-padding .. length + padding - 1) of each env's own clip -
 * into context_feat / context_mask, nothing else touched.  The reference's player calls it every context_length steps with the
 * current clip times (players/im_player.py:238-240): an evaluation rollout runs past the 32-step window without a reset.
 * V2P_ERR_INVALID when the env was created without a context buffer. */
int v2p_env_context(v2p_env* e, const int64_t* env_ids, int64_t n, const float* motion_times, void* stream);

/* BaseTask.step(actions) (base_task.py:147-165) = pre_physics_step + _physics_step +
 * post_physics_step.  actions [N,75] is masked IN PLACE for envs whose reset flag is set
 * (humanoid_smpl_im.py:126). */
int v2p_env_step(v2p_env* e, float* actions, void* stream);
/* The stages separately (trace replay / profiling). */
int v2p_env_pre_physics(v2p_env* e, float* actions, void* stream);   /* humanoid_smpl_im.py:125-157 */
int v2p_env_physics(v2p_env* e, void* stream);                        /* base_task.py:450-454: the 2 x gym.simulate */
int v2p_env_export(v2p_env* e, void* stream);                         /* the 6 gym.refresh_*_tensor calls, humanoid_smpl_im.py:452-468 */
int v2p_env_post_physics(v2p_env* e, void* stream);                   /* humanoid_smpl_im.py:398-418 */

/* set_actor_root_state_tensor_indexed + set_dof_state_tensor_indexed (+ rigid-body state for
 * teacher-forced replay): pushes the caller-edited root_states/dof_state (and rb_state when
 * `with_rb_state`) buffers into the engine's internal structure-of-arrays state. */
int v2p_env_push_state(v2p_env* e, const int64_t* env_ids, int64_t n, int with_rb_state, void* stream);

/* Two schedules of the same physics model are built: 0 = one LINK per lane (default: 32 lanes per env, state in
 * registers, level-synchronous tree recursions), 1 = one ENV per lane (LDS-resident workspace).  They agree to float32
 * rounding; the second one is kept as an independent cross-check (tests) and for A/B profiling. */
int v2p_env_set_schedule(v2p_env* e, int schedule);

/* index (0/1) of the CURRENT target inside v2p_env_buffers.target; the other one is the previous target */
int v2p_env_target_index(const v2p_env* e);

/* which build of the link-per-lane kernel the batch runs (v2p_sim_cfg.kernel_build): 1 = LDS-parked / 3 waves per SIMD, 2 = registers / 2 */
int v2p_env_kernel_build(const v2p_env* e);

/* diagnostics for tests: contact vertex ids chosen in the last substep, [N,24,4] int32, body*64+vertex or -1
 * (needs v2p_sim_cfg.debug_contacts >= 1) */
int v2p_env_debug_contacts(v2p_env* e, int32_t* out, void* stream);

/* the same for every substep of the last control step, [N,substeps*control_freq_inv,24,4] (needs v2p_sim_cfg.debug_contacts == 2) */
int v2p_env_debug_contacts_substeps(v2p_env* e, int32_t* out, void* stream);

/* diagnostics for tests: the wave-slot -> env order the NEXT physics launch will look up (`perm`, [N] int32, written out here from the
 * tables the last launch left; envs are handed to waves in descending order of their contact load, see DESIGN.md "pairing") and the
 * load key it is built from (`key`, [N]) */
int v2p_env_debug_pairing(v2p_env* e, int32_t* perm, int32_t* key, void* stream);

/* ---- racket + ball (vid2player/env/tasks/humanoid_smpl_im_mvae.py:367-442 actors, :711-783 physics step; data/assets/
 * smpl_mesh_humanoid_djokovic.xml:188-190, tennis_ball.urdf).  The racket is welded to a link: its mass belongs to that link of the
 * body model handed to v2p_model_create (vid2player3d_amd/racket.py builds it), its two solid cylinders are given here in the link's
 * frame for the ball contacts.  The ball is a free sphere simulated alongside the humanoid (same substeps): gravity, the reference's
 * drag + Magnus force re-evaluated before every simulate() call (apply_external_force_to_ball), ball x ground and ball x racket
 * contacts with restitution and friction (material values combined by averaging, PhysX's default), and - body_contacts - the ball
 * against the convex hulls of the humanoid's links (the ball actor collides with every shape of its env, :367-372, 432): one point per
 * substep, against the nearest hull; the racket's link is left to its cylinders.  Link-per-lane schedule, contacts on, either solver
 * (TGS: the two-body rows' gaps advance slice by slice; a row's restitution target is taken once, at the start of the substep).
 * The reference's flag bookkeeping around simulate() (bounce test on the ball height at the start of every call, :731-737; racket-hit
 * poll after it, :773-779) runs inside the step when the flag buffers are given. */
typedef struct {
    float radius, mass, inertia;                     /* 0.032, 0.057, 4e-5 */
    float restitution_ground, friction_ground;       /* 0.5, 0.9 */
    float restitution_racket, friction_racket;       /* 1.0, 0.8 */
    float bounce_threshold_velocity;                 /* 0.2 (sim.physx) */
    float angular_damping, max_angular_velocity;     /* gymapi.AssetOptions defaults: 0.5, 64 */
    float spin_scale;                                /* cfg_v2p spin_scale, 1.0 */
    int32_t racket_link;                             /* 22 = R_Wrist */
    int32_t num_cylinders;                           /* <= 2 */
    float cylinders[2][8];                           /* centre 3, unit axis 3, half length, radius; racket_link's frame */
    float racket_offset[3];                          /* origin of the racket rigid body (index 24 of the reference's tensor) in that frame */
    float restitution_body, friction_body;           /* ball x a link's hull: 0.5, 0.9 */
    int32_t body_contacts;                           /* 1: ball x hull contacts on */
    float bounce_height;                             /* ball height at the start of a simulate() call at or below which the ball "has bounced" (:733) */
    int32_t poll_racket_hits;                        /* 1: the racket-hit flags are kept (the reference does so when sim.substeps <= 2, :769) */
} v2p_ball_cfg;
typedef struct {                  /* caller-owned DEVICE buffers */
    float* ball_state;            /* [N,13] the ball actor's root state (pos quat linvel angvel): read at the start of every step, written at
                                   * its end - edit it in place to launch a ball (set_actor_root_state_tensor_indexed) */
    float* racket_state;          /* [N,13] rigid-body state of the racket */
    float* ball_per_sim;          /* [N,control_freq_inv,13] ball state after each simulate() call (what refresh_actor_root_state_tensor shows) */
    int32_t* racket_hit_per_sim;  /* [N,control_freq_inv] racket-ball contact force non-zero after the call (the reference's poll, :773-779) */
    float* ball_contact;          /* [N,2,3] contact force on the ball from the racket / from the ground after the last call */
    float* ball_body_contact;     /* [N,3] nullable: contact force on the ball from the humanoid's links after the last call */
    /* the reference's flags, all five or none (nullable): sticky `has_*` (cleared by the caller when it relaunches a ball), `*_now` = set by
     * this step; bounce_pos = ball position at the start of the call that saw the bounce */
    uint8_t* has_bounce;          /* [N] */
    uint8_t* has_bounce_now;      /* [N] */
    float* bounce_pos;            /* [N,3] */
    uint8_t* has_racket_contact;      /* [N] */
    uint8_t* has_racket_contact_now;  /* [N] */
    /* ---- ABI 9 */
    float* contact_force_sum;     /* [N,24,3] nullable: `_contact_forces_sum` (humanoid_smpl_im_mvae.py:186, 690, 781) - the net contact force of
                                   * every link (the racket's link carries the reaction of the ball) after each simulate() call, summed over the
                                   * calls of the control step; overwritten by every step */
} v2p_ball_buffers;
int v2p_env_attach_ball(v2p_env* e, const v2p_ball_cfg* cfg, const v2p_ball_buffers* buffers);

/* measurement: HIP events around every launch of the physics kernel (the dominant kernel of the step), recorded on the launch stream
 * by v2p_env_step / v2p_env_physics between _begin and _end (at most max_launches of them).  _end synchronises the events and returns
 * the summed kernel time in milliseconds and the number of launches measured.  bench.py's roofline.kernel_ms comes from here, from
 * the very steps it times. */
int v2p_env_profile_begin(v2p_env* e, int64_t max_launches);
/* the same around a SAMPLE of the launches (ABI 10): launch L since _begin is bracketed when L % stride == (L / period) % stride, so
 * that with period = the steps of an epoch every position of the epoch is measured once in `stride` epochs.  Two event records per
 * launch cost ~8 us of dispatch on this stack (2 % of a 0.41 ms step): bench.py measures one launch in eight. */
int v2p_env_profile_begin_sampled(v2p_env* e, int64_t max_launches, int32_t stride, int32_t period);
int v2p_env_profile_end(v2p_env* e, double* physics_ms_total, int64_t* launches);

/* Substep jobs: a job whose predecessor (the previous substep of its env pair, another workgroup of the launch) does not show up within
 * ~20 ms stops waiting and recomputes the pair's earlier substeps itself - results and progress do not depend on the order in which the
 * hardware dispatches workgroups, only time is lost.  Every such recovery is counted on the device (it should never happen: jobs are
 * numbered the way workgroups are dispatched).  v2p_env_check synchronises `stream` and fetches the counter; v2p_env_check_async (ABI 9)
 * enqueues the fetch behind the work already on `stream` and picks up what the previous call's fetch brought back, without waiting (one
 * call per epoch: HumanoidSMPLIM.reset of all envs makes it); v2p_env_job_recoveries returns the count as last fetched (total since
 * the batch was created).
 * A recovery loses time only.  One consequence of it can lose DATA: when the late predecessor finally starts after the whole step of its pair
 * is complete, it must not run (its inputs are the next step's) and is skipped - and what only that job publishes (the exposed PD targets,
 * the in-place masking of dead envs' actions, the ball's per-simulate() records) is then missing for that step.  Such skips are counted
 * too (v2p_env_jobs_skipped, ABI 13); v2p_env_check returns V2P_ERR_INTERNAL the first time it sees new ones. */
int v2p_env_check(v2p_env* e, void* stream);
int v2p_env_check_async(v2p_env* e, void* stream);
int v2p_env_job_recoveries(v2p_env* e, int64_t* count);
int v2p_env_jobs_skipped(v2p_env* e, int64_t* count);

const char* v2p_last_error(void);
int v2p_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* V2P_ROLLOUT_H */
