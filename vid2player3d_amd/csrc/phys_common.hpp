// Declarations shared by the two physics kernels (physics.hip: one env per lane, LDS resident;
// physics_ll.hip: one link per lane, register resident).
#pragma once
#include "phys_math.hpp"
#include "v2p_internal.hpp"

namespace v2p {

// ---------------------------------------------------------------------------- global workspace layout
// per-link slots of the global (structure-of-arrays [slot][env]) workspace: contact records + Lambda_b
constexpr int GL = 0;     // 21 Lambda_b: La(6) Lb(9) Lc(6)
constexpr int GCN = 21;   // 1  number of contacts
constexpr int GCR = 22;   // 12 contact offsets from the body origin
constexpr int GCB = 34;   // 4  contact bias
constexpr int GCL = 38;   // 12 contact impulses (n, t1, t2)
constexpr int LINK_SLOTS = 50;
constexpr int WS_SLOTS = LINK_SLOTS * NB;


// racket + ball (v2p_env_attach_ball): parameters by value, buffers borrowed from the caller
struct BallDev {
    float radius, mass, inv_mass, inv_inertia, rest_ground, fric_ground, rest_racket, fric_racket, bounce_thr, ang_damp, max_ang_vel, spin_scale;
    int32_t racket_link, ncyl, sub_per_sim, enabled;
    float cyl[2][8];       // centre 3, unit axis 3, half length, radius; in the racket link's frame
    float racket_off[3];   // origin of the exported racket rigid body in the link's frame
    float* state;          // [N,13] ball root state (pos quat vel angvel), read at the start of a step, written at its end
    float* racket_state;   // [N,13] rigid-body state of the racket (rigid body 24 of the reference's tensor)
    float* per_sim;        // [N,nsim,13] ball state after each simulate() call
    int32_t* hit_per_sim;  // [N,nsim] racket-ball contact force non-zero in the call's last substep
    float* contact;        // [N,2,3] force on the ball from the racket / from the ground, last substep
    float rest_body, fric_body, bounce_height;
    int32_t body_contacts, poll_hits;
    float* body_contact;   // [N,3] or NULL: force on the ball from the humanoid's links, last substep
    uint8_t *has_bounce, *has_bounce_now, *has_hit, *has_hit_now;  // the reference's flags (all or none)
    float* bounce_pos;     // [N,3]
    float* contact_sum;    // [N,24,3] or NULL: net contact forces of the links summed over the simulate() calls of the control step
    float* contact_part;   // [N,nsim,24,3] engine-owned (with contact_sum): the net contact forces after each simulate() call, one slot per call
};

// post-physics fused into the physics launch (v2p_env_step, link-per-lane schedule): what env_post_kernel takes
struct PostArgs {
    v2p_env_buffers b;
    v2p_motion_tables t;
    const int64_t* motion_id;
    int32_t cur;  // index of the current target buffer
    int32_t on;   // 1: the job of an env's last substep also runs its post-physics (obs, reward, reset flags, next target)
};

struct PhysArgs {
    const DevModel* __restrict__ model;
    float* __restrict__ state;
    float* ctrl;  // [N][75] PD targets + root wrench (written by pre-physics: its own kernel, or this kernel's prologue when `actions` is set)
    float* __restrict__ out;
    float* __restrict__ ws;
    int32_t* __restrict__ contact_ids;
    int32_t* __restrict__ contact_ids_sub;  // [N,nsub,24,4] or NULL
    // the caller's row-major tensors (link-per-lane kernel writes them directly: lane = body gives contiguous rows)
    float* __restrict__ x_root;     // [N,13]
    float* __restrict__ x_dof;      // [N,69,2]
    float* __restrict__ x_rb;       // [N,24,13]
    float* __restrict__ x_contact;  // [N,24,3]
    float* __restrict__ x_dof_force;  // [N,69]
    // fused pre-physics (v2p_env_step, link-per-lane schedule): the caller's actions [N,75] (masked in place), reset flags, exposed PD targets
    float* actions;
    const int64_t* reset;
    float* pd_target;
    long long* prof;  // optional cycle counters per phase (block 0), NULL = off
    int prof_heavy;   // sample the first 8 workgroups (the heaviest env pairs) instead of every 64th
    long long* wave_times;  // optional [waves][4]: wall-clock start, end (100 MHz), slot key, hw id of every wave of the launch
    const DevShape* shapes;    // [num_shapes] per-env body shapes (multi-shape batches only)
    const int32_t* env_shape;  // [N] shape of each env
    const float* shape_aug;    // [num_shapes][24] joint-diagonal augmentation of each shape
    // wave slot -> env (envs are handed to waves by contact load): slot -> rank (the heavy x light mix) -> load bin (pl_start) -> env (pl_list);
    // pl_start NULL = identity.  The launch fills pl_list_next / pair_start for the next one.
    const int32_t* pl_start;   // [256] first rank of each load bin (0 = heaviest)
    const int32_t* pl_list;    // [256][N] envs of each bin in arrival order
    int32_t* pl_list_next;     // [256][N]
    int32_t pl_mix;            // the `mix` heaviest envs are paired with the `mix` lightest ones
    int32_t* pl_slot_env;      // [N] env of each wave slot, handed from the job of a pair's first substep to the later ones
    int32_t* pair_hist;   // [256] envs per load bin (0 = heaviest), filled by atomics, consumed + cleared by the last workgroup
    int32_t* pair_start;  // [256] first slot of each bin (exclusive scan of the histogram), written by the last workgroup
    int32_t* pair_done;   // [1] workgroups that have finished
    int32_t* pair_pos;    // [N] arrival index of the env inside its bin
    int32_t* pair_key;    // [N] load key of the env (touched links of the last substep x 8 + depth of the deepest), 0..255
    unsigned long long par_pack[2];  // parents[24] and level order[24], 5 bits each, 12 per word: the tree walks of the
    unsigned long long ord_pack[2];  // impulse sweep decode them with scalar ALU ops instead of dependent scalar loads
    int64_t n;
    EnvParams p;
    BallDev ball;
    // substep jobs (physics_ll.hip JOBS): workgroups per substep, progress word per wave slot (+1 error word), epoch of this launch
    int32_t job_blocks, job_epoch, job_mono;
    int32_t* job_progress;
    float* job_hand;  // [nsub - 1][N][HAND_FLOATS] state hand-off between the substep jobs of an env pair: slot s = the state after substep s
    int32_t job_len;         // substeps per job
    int32_t job_lead;        // ... of the first job of a cut pair (>= 1)
    int32_t job_interleave;  // 1: heavy x light slots and pairs of equals are dispatched alternately (see the kernel)
    long job_timeout_spins;  // polls (of ~0.4 us) after which a job stops waiting for its predecessor and recomputes the earlier substeps
    PostArgs post;
};


// ---- pairing (see physics_ll.hip): slot of env e in the next launch = first slot of its load bin + its arrival index there
constexpr int PAIR_BINS = 256;
struct PairView {
    const int32_t* key;    // [N]
    const int32_t* pos;    // [N]
    const int32_t* start;  // [PAIR_BINS]
    int32_t* perm;         // [N], NULL = nothing to scatter
    int32_t n, mix;        // envs; the `mix` heaviest envs are paired with the `mix` lightest ones instead of with each other
};
// slot of env e in the next launch.  r = rank of e by contact load, descending (first slot of its load bin + its arrival index).
// Envs are paired by rank, (0,1), (2,3), ...: a wave costs the UNION of its two envs' group structure, pairing equals keeps that
// union small.  Except for the `mix` heaviest: their chain of block updates is the critical path of the whole launch, and next to an
// equally heavy but differently shaped partner it grows by the partner's irregularities - each of them shares its wave with one of
// the `mix` lightest envs instead (rank r < mix: slot 2r; rank r >= n - mix: slot 2 (n-1-r) + 1; the rest follows in rank order).
__device__ __forceinline__ void pair_scatter(const PairView& pv, int64_t e) {
    const int r = pv.start[PAIR_BINS - 1 - pv.key[e]] + pv.pos[e];
    int slot = r;
    if (pv.mix > 0) {
        if (r < pv.mix) slot = 2 * r;
        else if (r >= pv.n - pv.mix) slot = 2 * (pv.n - 1 - r) + 1;
        else slot = r + pv.mix;
    }
    pv.perm[slot] = (int32_t)e;
}

}  // namespace v2p
