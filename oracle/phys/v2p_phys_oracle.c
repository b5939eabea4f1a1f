/*
 * v2p_phys_oracle.c -- CPU restatement (plain C, float64, dense linear algebra) of the
 * articulated rigid-body step that replaces `gym.simulate` for the SMPL humanoid.
 *
 * TEST INFRASTRUCTURE ONLY.  Linked/loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker; the product (vid2player3d_amd/) never
 * touches it.
 *
 * PARITY UNPINNED.  The reference delegates this step to Isaac Gym Preview 4 / PhysX 5 GPU
 * (closed binary, absent from /root/reference; call sites embodied_pose/env/tasks/
 * base_task.py:450-454, humanoid_smpl_im.py:134,154,453-467; parameters
 * embodied_pose/cfg/amass_im.yaml:32-52, embodied_pose/utils/config.py:190-222; actor
 * setup humanoid_smpl_im.py:273-276,356-389).  No golden vectors exist for it and none can
 * be produced here.  What this file pins instead is the *published algorithm class* PhysX
 * implements -- reduced-coordinate articulation (floating base + spherical joints), implicit
 * PD joint drives, convex-hull-vertex vs ground-plane contact with a <=4 point manifold per
 * body, projected Gauss-Seidel contact/friction rows, semi-implicit Euler -- written the
 * slow, obviously-correct way (body Jacobians, dense mass matrix, Cholesky) so that the HIP
 * kernels, which use an O(n) articulated-body recursion, can be checked against an
 * independent formulation.  Physical invariants (energy, momentum, rest contact) are checked
 * on top in tests/test_phys_oracle.py.
 *
 * Model ("v2p physics v1"), one substep of length h:
 *   generalized velocity  v = [xdot_0 (world), w_0 (world), wrel_b (b=1..B-1, body-b axes)]
 *   Mt = M(q) + diag_joint(armature + h*kd + h^2*kp)
 *   v* = v + h * Mt^-1 ( Q_ext - C(q,v) + kp*(q_tar - q) - (kd + h*kp)*wrel )
 *        q = exponential-map coordinates of the joint quaternion (the reference's dof_pos
 *        convention, embodied_pose/utils/motion_lib.py:460-488)
 *   contacts: hull vertices with z < contact_offset; <=4 per body (deepest, farthest,
 *        extreme left/right); rows n,t1,t2 per point; bias = d/h (d>=0) or
 *        max(erp*d/h, -max_depenetration_velocity) (d<0); box friction |lt| <= mu*ln
 *   PGS (solver_type 0): n_iter sweeps, bodies ascending; inside a body: limit rows of its joint, hull points in slot order, rows
 *        n,t1,t2.  (v2p_oracle_experiment(4): sweeps in alternating direction - an experiment of round 4, see the sweep.)
 *   TGS (solver_type 1): one sweep per time slice h/n_iter with re-evaluated gaps (see the substep)
 *   v+ = v* + Mt^-1 J^T lambda;  angular damping 1/(1+h*c); |w| clamp; integrate.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NB 24
#define NJ (NB - 1)
#define ND (6 + 3 * NJ)
#define MAXC_BODY 4
#define MAXROWS (NB * MAXC_BODY * 3)
#define MAXV_BODY 64 /* hull vertices per body (the engine's limit as well) */

typedef struct {
    int parents[NB];
    double local_pos[NB][3];
    double mass[NB];
    double com[NB][3];
    double inertia[NB][9]; /* about COM, body axes */
    double kp[3 * NJ], kd[3 * NJ], armature[3 * NJ];
    int hull_offsets[NB + 1];
    const double *hull_verts; /* [V][3] body frame */
    double limit_lo[3 * NJ], limit_hi[3 * NJ]; /* per-DOF range of the exponential-map coordinate, radians (MJCF `range`) */
} v2p_omodel;

typedef struct {
    double h;
    double gravity_z;      /* -9.81 */
    double mu;             /* 1.0 */
    double contact_offset; /* 0.02 */
    double max_depen_vel;  /* 10.0 */
    double ang_damp;       /* 0.01 */
    double max_ang_vel;    /* 100.0 */
    double erp;            /* 0.2: fraction of a penetration corrected per substep */
    int n_iter;            /* 4 */
    int enable_contact;
    int solver_type;       /* 0 = PGS (default), 1 = TGS (see the substep) */
    int joint_limits;      /* 1: DOFs whose range is narrower than a full turn get a limit row (see the substep) */
    double limit_margin;   /* 0.05 rad: the row exists only while C < limit_margin + h max(0, approach rate of v*) */
    double rest_offset;    /* 0.0 (sim.physx.rest_offset): the gap of a hull-vertex row is z - rest_offset */
    int friction_frame;    /* 0 = world (t1 = x, t2 = y: the friction limit of a hull x ground point is a box aligned with the world axes);
                            * 1 = velocity: t1 along the tangential velocity the contact point has under v* (the unconstrained velocity of the
                            * substep: where the point would slide without contact impulses), t2 = n x t1; below 1e-6 m/s the world frame.
                            * PhysX aligns its friction directions with the relative velocity at the contact; which of the two is closer to
                            * Isaac Gym is for a trace to say (tools/replay_trace.py --friction-frame).  Hull x ground rows only: the ball's
                            * rows keep the basis of their normal. */
} v2p_oparams;

/* ---- racket + ball (vid2player/env/tasks/humanoid_smpl_im_mvae.py:367-442, 711-783; data/assets/tennis_ball.urdf,
 * smpl_mesh_humanoid_djokovic.xml:188-190).  The racket is welded to a link of the articulation (its mass is part of that link's
 * body model, its two cylinders are given here in the link's frame); the ball is a free sphere.  Modelled contacts: ball-ground,
 * ball-racket (sphere against the two solid cylinders) and - body_contacts - ball against the convex hulls of the humanoid's links
 * (the ball actor's collision filter is 0, the humanoid's 1: every pair collides, humanoid_smpl_im_mvae.py:367-372, 432): ONE point per
 * substep, against the link whose hull is nearest (the racket's link is left to its cylinders). */
typedef struct { double center[3], axis[3], half_len, radius; } v2p_ocyl;
typedef struct {
    double radius, mass, inertia;   /* 0.032, 0.057, 4e-5 (tennis_ball.urdf) */
    double rest_ground, fric_ground; /* ball x plane, PhysX default combine = average: (1.0 + 0.0) / 2, (0.8 + 1.0) / 2 */
    double rest_racket, fric_racket; /* ball x racket head: 1.0, 0.8 (humanoid_smpl_im_mvae.py:414-416, 436-438) */
    double rest_body, fric_body;     /* ball x a link's hull (shape defaults 0 / 1 on the humanoid's side): (1.0 + 0.0) / 2, (0.8 + 1.0) / 2 */
    double bounce_threshold;         /* 0.2 m/s (sim.physx.bounce_threshold_velocity) */
    double ang_damp, max_ang_vel;    /* AssetOptions defaults of the ball asset: 0.5, 64 */
    int racket_link;                 /* 22 = R_Wrist */
    int ncyl;
    int body_contacts;               /* 1: ball x hull contacts on */
    v2p_ocyl cyl[2];
} v2p_oball_params;
typedef struct { double pos[3], quat[4], vel[3], angvel[3]; } v2p_oball;

/* optional per-substep inputs / outputs for the parity tests (all nullable) */
typedef struct {
    const int *forced_ids; /* [NB*4] in: use these hull vertices (body*64+vertex, -1 = none) instead of the selection rule */
    int *own_ids;          /* [NB*4] out: what the selection rule picks in this state (whether or not it was forced) */
    double *margins;       /* [NB] out: how close the selection rule came to deciding otherwise, in metres (1e30 = no decision) */
    double *clamp_margin;  /* [1] in/out (min): how close any row update of the sweep came to the other side of its clamp, as the change of
                            * the row's relative velocity (m/s, rad/s for a limit row) that would have switched it: |unclamped impulse -
                            * switching value| * w_ii.  A float32 evaluation of the same sweep can take the other branch when this is
                            * of the order of its rounding; the parity tests attribute their outliers with it. */
} v2p_osub_io;

typedef struct {
    double root_pos[3];
    double root_quat[4];  /* xyzw */
    double jquat[NJ][4];  /* parent->child, xyzw */
    double vel[ND];
} v2p_ostate;

int v2p_oracle_substep_io(const v2p_omodel *m, const v2p_oparams *p, v2p_ostate *s, const double *pd_target, const double *ext_force,
                          const double *ext_torque, double *contact_force, double *dof_force, int *contact_ids, const v2p_osub_io *io);

static __thread int g_hull_rows; /* ball x hull points of the last substep (diagnostics of the parity tests) */

/* ------------------------------------------------------------------ small helpers */
static void cross(const double a[3], const double b[3], double o[3]) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void qmul(const double a[4], const double b[4], double o[4]) {
    double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
static void qnormalize(double q[4]) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
}
static void q2mat(const double q[4], double R[9]) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
static void matvec(const double R[9], const double v[3], double o[3]) {
    double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    double y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    double z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
/* rotation vector -> quaternion */
static void rotvec2quat(const double v[3], double q[4]) {
    double a = sqrt(dot3(v, v));
    double k = a > 1e-12 ? sin(0.5 * a) / a : 0.5;
    q[0] = v[0] * k; q[1] = v[1] * k; q[2] = v[2] * k; q[3] = cos(0.5 * a);
}
/* quaternion -> exponential map, the reference's convention
 * (embodied_pose/utils/torch_utils.py:82-119): angle = wrap(2 acos w), axis = xyz/sqrt(1-w^2),
 * default axis z and angle 0 when |sin| <= 1e-5. */
static void quat2expmap(const double q[4], double e[3]) {
    double w = q[3];
    double s2 = 1.0 - w * w;
    double s = s2 > 0 ? sqrt(s2) : 0.0;
    if (!(fabs(s) > 1e-5)) { e[0] = e[1] = e[2] = 0.0; return; }
    double wc = w > 1 ? 1 : (w < -1 ? -1 : w);
    double ang = 2.0 * acos(wc);
    ang = atan2(sin(ang), cos(ang));
    e[0] = ang * q[0] / s; e[1] = ang * q[1] / s; e[2] = ang * q[2] / s;
}
/* exponential map -> quaternion (torch_utils.py:144-166) */
static void expmap2quat(const double e[3], double q[4]) {
    double ang = sqrt(dot3(e, e));
    double ax[3] = {0, 0, 1};
    double angn = atan2(sin(ang), cos(ang));
    if (fabs(angn) > 1e-5) { ax[0] = e[0] / ang; ax[1] = e[1] / ang; ax[2] = e[2] / ang; } else { angn = 0.0; }
    double sh = sin(0.5 * angn);
    q[0] = ax[0] * sh; q[1] = ax[1] * sh; q[2] = ax[2] * sh; q[3] = cos(0.5 * angn);
    qnormalize(q);
}

/* ------------------------------------------------------------------ kinematics */
typedef struct {
    double R[NB][9];   /* world <- body */
    double x[NB][3];   /* body origin, world */
    double w[NB][3];   /* angular velocity, world */
    double xd[NB][3];  /* origin linear velocity, world */
    double quat[NB][4];
} kin_t;

static void kinematics(const v2p_omodel *m, const v2p_ostate *s, kin_t *k) {
    for (int b = 0; b < NB; ++b) {
        int p = m->parents[b];
        if (p < 0) {
            memcpy(k->quat[b], s->root_quat, sizeof(double) * 4);
            memcpy(k->x[b], s->root_pos, sizeof(double) * 3);
            q2mat(k->quat[b], k->R[b]);
            for (int i = 0; i < 3; ++i) { k->xd[b][i] = s->vel[i]; k->w[b][i] = s->vel[3 + i]; }
        } else {
            qmul(k->quat[p], s->jquat[b - 1], k->quat[b]);
            qnormalize(k->quat[b]);
            q2mat(k->quat[b], k->R[b]);
            double r[3], wr[3], t[3];
            matvec(k->R[p], m->local_pos[b], r);
            for (int i = 0; i < 3; ++i) k->x[b][i] = k->x[p][i] + r[i];
            matvec(k->R[b], &s->vel[6 + 3 * (b - 1)], wr);
            for (int i = 0; i < 3; ++i) k->w[b][i] = k->w[p][i] + wr[i];
            cross(k->w[p], r, t);
            for (int i = 0; i < 3; ++i) k->xd[b][i] = k->xd[p][i] + t[i];
        }
    }
}

/* body Jacobian: [w_b; xdot_b] = J_b v, J_b is 6 x ND (row-major) */
static void body_jacobian(const v2p_omodel *m, const kin_t *k, int b, double *J) {
    memset(J, 0, sizeof(double) * 6 * ND);
    /* root linear */
    for (int i = 0; i < 3; ++i) J[(3 + i) * ND + i] = 1.0;
    /* root angular */
    double r0[3] = {k->x[b][0] - k->x[0][0], k->x[b][1] - k->x[0][1], k->x[b][2] - k->x[0][2]};
    for (int i = 0; i < 3; ++i) {
        double e[3] = {0, 0, 0}, c[3];
        e[i] = 1.0;
        J[i * ND + 3 + i] = 1.0;
        cross(e, r0, c);
        for (int r = 0; r < 3; ++r) J[(3 + r) * ND + 3 + i] = c[r];
    }
    for (int a = b; a > 0; a = m->parents[a]) {
        double ra[3] = {k->x[b][0] - k->x[a][0], k->x[b][1] - k->x[a][1], k->x[b][2] - k->x[a][2]};
        for (int i = 0; i < 3; ++i) {
            double ax[3] = {k->R[a][0 * 3 + i], k->R[a][1 * 3 + i], k->R[a][2 * 3 + i]}, c[3];
            int col = 6 + 3 * (a - 1) + i;
            for (int r = 0; r < 3; ++r) J[r * ND + col] = ax[r];
            cross(ax, ra, c);
            for (int r = 0; r < 3; ++r) J[(3 + r) * ND + col] = c[r];
        }
    }
}

/* spatial inertia at the body origin, world axes: [A B; B^T m1] */
static void spatial_inertia(const v2p_omodel *m, const kin_t *k, int b, double I6[36], double d[3], double Ic[9]) {
    const double *R = k->R[b];
    double tmp[9];
    /* Ic = R I R^T */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int l = 0; l < 3; ++l) s += R[i * 3 + l] * m->inertia[b][l * 3 + j];
            tmp[i * 3 + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int l = 0; l < 3; ++l) s += tmp[i * 3 + l] * R[j * 3 + l];
            Ic[i * 3 + j] = s;
        }
    matvec(R, m->com[b], d);
    double ms = m->mass[b];
    double dx[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0}; /* [d]x */
    memset(I6, 0, sizeof(double) * 36);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double dd = 0; /* [d]x [d]x */
            for (int l = 0; l < 3; ++l) dd += dx[i * 3 + l] * dx[l * 3 + j];
            I6[i * 6 + j] = Ic[i * 3 + j] - ms * dd;
            I6[i * 6 + 3 + j] = ms * dx[i * 3 + j];
            I6[(3 + i) * 6 + j] = -ms * dx[i * 3 + j];
        }
    for (int i = 0; i < 3; ++i) I6[(3 + i) * 6 + 3 + i] = ms;
}

/* dense Cholesky (lower) in place; returns 0 on success */
static int cholesky(double *A, int n) {
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
        if (s <= 0) return -1;
        double l = sqrt(s);
        A[j * n + j] = l;
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (int k = 0; k < j; ++k) t -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = t / l;
        }
    }
    return 0;
}
static void chol_solve(const double *L, int n, double *b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k];
        b[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
}

/* ------------------------------------------------------------------ dynamics terms */
/* M (ND x ND) and bias C (Coriolis/centrifugal + gravity) via body Jacobians. */
static void mass_and_bias(const v2p_omodel *m, const v2p_oparams *p, const v2p_ostate *s, const kin_t *k, double *M, double *C) {
    double J[6 * ND];
    double avp[NB][3], xvp[NB][3]; /* velocity-product accelerations (vdot = 0) */
    memset(M, 0, sizeof(double) * ND * ND);
    memset(C, 0, sizeof(double) * ND);
    for (int b = 0; b < NB; ++b) {
        int par = m->parents[b];
        if (par < 0) {
            for (int i = 0; i < 3; ++i) avp[b][i] = xvp[b][i] = 0.0;
        } else {
            double wr[3], t[3], r[3], t2[3];
            matvec(k->R[b], &s->vel[6 + 3 * (b - 1)], wr);
            cross(k->w[par], wr, t);
            for (int i = 0; i < 3; ++i) avp[b][i] = avp[par][i] + t[i];
            matvec(k->R[par], m->local_pos[b], r);
            cross(avp[par], r, t);
            cross(k->w[par], r, t2);
            cross(k->w[par], t2, t2);
            for (int i = 0; i < 3; ++i) xvp[b][i] = xvp[par][i] + t[i] + t2[i];
        }
        double I6[36], d[3], Ic[9];
        spatial_inertia(m, k, b, I6, d, Ic);
        body_jacobian(m, k, b, J);
        /* M += J^T I6 J */
        double IJ[6 * ND];
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < ND; ++c) {
                double sum = 0;
                for (int l = 0; l < 6; ++l) sum += I6[r * 6 + l] * J[l * ND + c];
                IJ[r * ND + c] = sum;
            }
        for (int r = 0; r < ND; ++r)
            for (int c = 0; c < ND; ++c) {
                double sum = 0;
                for (int l = 0; l < 6; ++l) sum += J[l * ND + r] * IJ[l * ND + c];
                M[r * ND + c] += sum;
            }
        /* force needed with vdot=0: f6 = I6 a_vp + velocity terms - gravity */
        double a6[6] = {avp[b][0], avp[b][1], avp[b][2], xvp[b][0], xvp[b][1], xvp[b][2]};
        double f6[6];
        for (int r = 0; r < 6; ++r) {
            double sum = 0;
            for (int l = 0; l < 6; ++l) sum += I6[r * 6 + l] * a6[l];
            f6[r] = sum;
        }
        double Iw[3], wIw[3], wd[3], wwd[3], t[3];
        const double *w = k->w[b];
        matvec(Ic, w, Iw);
        cross(w, Iw, wIw);
        cross(w, d, wd);
        cross(w, wd, wwd);
        double ms = m->mass[b];
        double g[3] = {0, 0, p->gravity_z};
        double fl[3] = {ms * (wwd[0] - g[0]), ms * (wwd[1] - g[1]), ms * (wwd[2] - g[2])};
        cross(d, fl, t);
        for (int i = 0; i < 3; ++i) { f6[i] += wIw[i] + t[i]; f6[3 + i] += fl[i]; }
        for (int c = 0; c < ND; ++c) {
            double sum = 0;
            for (int l = 0; l < 6; ++l) sum += J[l * ND + c] * f6[l];
            C[c] += sum;
        }
    }
}

/* ------------------------------------------------------------------ experiments (root-causing of parity outliers; never used by a test's pass criterion)
 * bit 0: the GAP of every hull-vertex contact row (the z of the vertex that enters the row's bias d / h) comes from a float32 forward
 *        kinematics of the same state (quaternion chain and positions in float, link by link), everything else stays float64.  This
 *        isolates one float32 effect: a position error of ~1e-7 m is multiplied by 1 / h = 120 (erp / h = 24 when penetrating) in the
 *        velocity target of the row. */
static int g_experiment;
static double g_experiment_param;
void v2p_oracle_experiment(int flags) { g_experiment = flags; }
void v2p_oracle_experiment_param(double x) { g_experiment_param = x; } /* bit 1: every hull-vertex gap is shifted by this many metres */
static double f32_vertex_z(const v2p_omodel *m, const v2p_ostate *s, int body, int vert) {
    float q[NB][4], x[NB][3];
    for (int b = 0; b < NB; ++b) {
        int p = m->parents[b];
        if (p < 0) {
            for (int i = 0; i < 4; ++i) q[b][i] = (float)s->root_quat[i];
            for (int i = 0; i < 3; ++i) x[b][i] = (float)s->root_pos[i];
        } else {
            const float *a = q[p];
            float c[4] = {(float)s->jquat[b - 1][0], (float)s->jquat[b - 1][1], (float)s->jquat[b - 1][2], (float)s->jquat[b - 1][3]};
            float o[4] = {a[3] * c[0] + a[0] * c[3] + a[1] * c[2] - a[2] * c[1], a[3] * c[1] + a[1] * c[3] + a[2] * c[0] - a[0] * c[2],
                          a[3] * c[2] + a[2] * c[3] + a[0] * c[1] - a[1] * c[0], a[3] * c[3] - a[0] * c[0] - a[1] * c[1] - a[2] * c[2]};
            float n = sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
            for (int i = 0; i < 4; ++i) q[b][i] = o[i] / n;
            /* x_b = x_p + R_p local_pos */
            float X = a[0], Y = a[1], Z = a[2], W = a[3];
            float l[3] = {(float)m->local_pos[b][0], (float)m->local_pos[b][1], (float)m->local_pos[b][2]};
            float R[9] = {1 - 2 * (Y * Y + Z * Z), 2 * (X * Y - Z * W), 2 * (X * Z + Y * W), 2 * (X * Y + Z * W), 1 - 2 * (X * X + Z * Z), 2 * (Y * Z - X * W),
                          2 * (X * Z - Y * W), 2 * (Y * Z + X * W), 1 - 2 * (X * X + Y * Y)};
            for (int i = 0; i < 3; ++i) x[b][i] = x[p][i] + (R[3 * i] * l[0] + R[3 * i + 1] * l[1] + R[3 * i + 2] * l[2]);
        }
    }
    const float *a = q[body];
    float X = a[0], Y = a[1], Z = a[2], W = a[3];
    const double *hv = &m->hull_verts[3 * (m->hull_offsets[body] + vert)];
    float l[3] = {(float)hv[0], (float)hv[1], (float)hv[2]};
    return (double)(x[body][2] + ((2 * (X * Z - Y * W)) * l[0] + (2 * (Y * Z + X * W)) * l[1] + (1 - 2 * (X * X + Y * Y)) * l[2]));
}

/* ------------------------------------------------------------------ contact generation */
typedef struct {
    int n;
    int body[NB * MAXC_BODY];
    int vert[NB * MAXC_BODY];
    double pos[NB * MAXC_BODY][3];
} contacts_t;

/* Selection rule of one body (<=4 of the hull vertices below the contact offset: all if <= 4, else the deepest, the one
 * farthest from it in the plane, and the extremes on either side of that line; ties go to the lowest index).  `margin` = how
 * close the rule came to deciding otherwise, in metres: distance of any vertex to the offset plane, gap between the winner
 * and the runner-up of every arg-extreme (areas divided by the base length). */
static int select_body(const double (*P)[3], int nv, double coff, int sel[4], double *margin) {
    int count = 0, first[4] = {-1, -1, -1, -1}, k0 = -1, ns = 0;
    double zmin = 0, mg = 1e30;
    for (int i = 0; i < nv; ++i) {
        double dz = fabs(P[i][2] - coff);
        if (dz < mg) mg = dz;
        if (P[i][2] < coff) {
            if (count < 4) first[count] = i;
            if (k0 < 0 || P[i][2] < zmin) { k0 = i; zmin = P[i][2]; }
            ++count;
        }
    }
    if (count > 0 && count <= 4) {
        for (int i = 0; i < count; ++i) sel[ns++] = first[i];
    } else if (count > 4) {
        int k1 = -1, k2 = -1, k3 = -1;
        double best = -1.0, second = -1.0;
        for (int i = 0; i < nv; ++i) {
            if (!(P[i][2] < coff) || i == k0) continue;
            double g = P[i][2] - zmin; /* runner-up of the deepest point */
            if (g < mg) mg = g;
            double dx = P[i][0] - P[k0][0], dy = P[i][1] - P[k0][1];
            double d2 = dx * dx + dy * dy;
            if (d2 > best) { second = best; best = d2; k1 = i; } else if (d2 > second) second = d2;
        }
        if (second >= 0 && sqrt(best) - sqrt(second) < mg) mg = sqrt(best) - sqrt(second);
        double ex = P[k1][0] - P[k0][0], ey = P[k1][1] - P[k0][1];
        double el = sqrt(ex * ex + ey * ey) + 1e-300;
        double amax = 0.0, amin = 0.0, amax2 = 0.0, amin2 = 0.0;
        for (int i = 0; i < nv; ++i) {
            if (!(P[i][2] < coff) || i == k0 || i == k1) continue;
            double area = ex * (P[i][1] - P[k0][1]) - ey * (P[i][0] - P[k0][0]);
            if (area > amax) { amax2 = amax; amax = area; k2 = i; } else if (area > amax2) amax2 = area;
            if (area < amin) { amin2 = amin; amin = area; k3 = i; } else if (area < amin2) amin2 = area;
        }
        if (k2 >= 0 && (amax - amax2) / el < mg) mg = (amax - amax2) / el;
        if (k3 >= 0 && (amin2 - amin) / el < mg) mg = (amin2 - amin) / el;
        sel[ns++] = k0;
        sel[ns++] = k1;
        if (k2 >= 0) sel[ns++] = k2;
        if (k3 >= 0) sel[ns++] = k3;
    }
    if (margin) *margin = mg;
    return ns;
}

static void gen_contacts(const v2p_omodel *m, const v2p_oparams *p, const kin_t *k, contacts_t *cs, const v2p_osub_io *io) {
    cs->n = 0;
    for (int b = 0; b < NB; ++b) {
        int v0 = m->hull_offsets[b], v1 = m->hull_offsets[b + 1];
        int nv = v1 - v0;
        double P[MAXV_BODY][3]; /* (no heap traffic inside the OpenMP batch loop) */
        if (nv > MAXV_BODY) nv = MAXV_BODY;
        for (int i = 0; i < nv; ++i) {
            double t[3];
            matvec(k->R[b], &m->hull_verts[3 * (v0 + i)], t);
            for (int c = 0; c < 3; ++c) P[i][c] = k->x[b][c] + t[c];
        }
        int sel[4], ns;
        double mg;
        ns = select_body((const double (*)[3])P, nv, p->contact_offset, sel, &mg);
        if (io && io->own_ids) for (int i = 0; i < 4; ++i) io->own_ids[4 * b + i] = i < ns ? b * 64 + sel[i] : -1;
        if (io && io->margins) io->margins[b] = mg;
        if (io && io->forced_ids) {
            ns = 0;
            for (int i = 0; i < 4; ++i) {
                int id = io->forced_ids[4 * b + i];
                if (id >= 0 && id / 64 == b && id % 64 < nv) sel[ns++] = id % 64;
            }
        }
        for (int i = 0; i < ns; ++i) {
            int c = cs->n++;
            cs->body[c] = b;
            cs->vert[c] = sel[i];
            memcpy(cs->pos[c], P[sel[i]], sizeof(double) * 3);
        }
    }
}

/* ------------------------------------------------------------------ one substep */
/* closest point of a solid cylinder (centre c, unit axis a, half length hl, radius rc) to the point s; returns the distance */
static double cyl_closest(const double c[3], const double a[3], double hl, double rc, const double s[3], double pt[3], double n[3]) {
    double d[3] = {s[0] - c[0], s[1] - c[1], s[2] - c[2]};
    double t = dot3(d, a);
    double q[3] = {d[0] - t * a[0], d[1] - t * a[1], d[2] - t * a[2]};
    double rho = sqrt(dot3(q, q));
    double tc = t > hl ? hl : (t < -hl ? -hl : t);
    double k = rho > rc ? rc / rho : 1.0;
    for (int i = 0; i < 3; ++i) pt[i] = c[i] + tc * a[i] + k * q[i];
    double e[3] = {s[0] - pt[0], s[1] - pt[1], s[2] - pt[2]};
    double dist = sqrt(dot3(e, e));
    if (dist > 1e-9) { for (int i = 0; i < 3; ++i) n[i] = e[i] / dist; }
    else { double sg = t >= 0 ? 1.0 : -1.0; for (int i = 0; i < 3; ++i) n[i] = sg * a[i]; } /* centre inside the solid: leave through the nearer cap */
    return dist;
}
/* tangent basis of a contact normal (the same rule in the kernel): t1 = normalize(n x z) unless n is within 1e-3 of +-z, then n x x */
static void tangent_basis(const double n[3], double t1[3], double t2[3]) {
    const double ez[3] = {0, 0, 1}, ex[3] = {1, 0, 0};
    cross(n, ez, t1);
    if (dot3(t1, t1) < 1e-6) cross(n, ex, t1);
    double l = sqrt(dot3(t1, t1));
    for (int i = 0; i < 3; ++i) t1[i] /= l;
    cross(n, t1, t2);
}

/* ---- closest point of the convex hull of a vertex list to a point (GJK on the points V_i - c with the closest-point-on-simplex
 * rules of Ericson, "Real-Time Collision Detection" 5.1): barycentric weights of the closest point of a segment / triangle /
 * tetrahedron to the origin.  tests/test_oracle_ball_hull.py checks it against a quadratic program solved by scipy. */
static void seg_weights(const double a[3], const double b[3], double w[2]) {
    double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    double den = dot3(ab, ab), t = den > 0 ? -dot3(a, ab) / den : 0.0;
    t = t < 0 ? 0 : (t > 1 ? 1 : t);
    w[0] = 1 - t; w[1] = t;
}
static void tri_weights(const double a[3], const double b[3], const double c[3], double w[3]) {
    double ab[3], ac[3];
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; }
    double d1 = -dot3(ab, a), d2 = -dot3(ac, a);
    w[0] = w[1] = w[2] = 0;
    if (d1 <= 0 && d2 <= 0) { w[0] = 1; return; }
    double d3 = -dot3(ab, b), d4 = -dot3(ac, b);
    if (d3 >= 0 && d4 <= d3) { w[1] = 1; return; }
    double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) { double v = d1 / (d1 - d3); w[0] = 1 - v; w[1] = v; return; }
    double d5 = -dot3(ab, c), d6 = -dot3(ac, c);
    if (d6 >= 0 && d5 <= d6) { w[2] = 1; return; }
    double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) { double u = d2 / (d2 - d6); w[0] = 1 - u; w[2] = u; return; }
    double va = d3 * d6 - d5 * d4;
    if (va <= 0 && d4 - d3 >= 0 && d5 - d6 >= 0) { double u = (d4 - d3) / ((d4 - d3) + (d5 - d6)); w[1] = 1 - u; w[2] = u; return; }
    double den = 1.0 / (va + vb + vc);
    w[1] = vb * den; w[2] = vc * den; w[0] = 1 - w[1] - w[2];
}
/* returns 1 when the origin lies inside the tetrahedron */
static int tet_weights(const double P[4][3], double w[4]) {
    static const int F[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}}; /* face, opposite vertex */
    double best = 1e300;
    int any = 0;
    for (int f = 0; f < 4; ++f) {
        const double *a = P[F[f][0]], *b = P[F[f][1]], *c = P[F[f][2]], *d = P[F[f][3]];
        double ab[3], ac[3], n[3], ad[3];
        for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ad[i] = d[i] - a[i]; }
        cross(ab, ac, n);
        double so = -dot3(n, a), sd = dot3(n, ad);
        if (so * sd < 0 || sd == 0) { /* the origin is on the far side of this face (or the tetrahedron is flat) */
            double t[3], q[3];
            tri_weights(a, b, c, t);
            for (int i = 0; i < 3; ++i) q[i] = t[0] * a[i] + t[1] * b[i] + t[2] * c[i];
            double dd = dot3(q, q);
            if (dd < best) { best = dd; any = 1; w[0] = w[1] = w[2] = w[3] = 0; w[F[f][0]] = t[0]; w[F[f][1]] = t[1]; w[F[f][2]] = t[2]; }
        }
    }
    return !any;
}
/* distance of c from the hull; p = the closest point (c itself when c is inside: distance 0) */
static double hull_closest(const double *V /*[nv][3]*/, int nv, const double c[3], double p[3]) {
    double S[4][3], v[3];
    int idx[4], n = 1;
    for (int i = 0; i < 3; ++i) v[i] = S[0][i] = V[i] - c[i];
    idx[0] = 0;
    for (int it = 0; it < 64; ++it) {
        int best = 0;
        double bd = 1e300;
        for (int k = 0; k < nv; ++k) {
            double d = v[0] * (V[3 * k] - c[0]) + v[1] * (V[3 * k + 1] - c[1]) + v[2] * (V[3 * k + 2] - c[2]);
            if (d < bd) { bd = d; best = k; }
        }
        double vv = dot3(v, v);
        if (vv - bd <= 1e-14 * vv + 1e-26) break; /* no vertex lies closer along -v: v is the closest point */
        int dup = 0;
        for (int k = 0; k < n; ++k) dup |= idx[k] == best;
        if (dup) break;
        for (int i = 0; i < 3; ++i) S[n][i] = V[3 * best + i] - c[i];
        idx[n++] = best;
        double w[4] = {0, 0, 0, 0};
        int inside = 0;
        if (n == 2) seg_weights(S[0], S[1], w);
        else if (n == 3) tri_weights(S[0], S[1], S[2], w);
        else inside = tet_weights((const double(*)[3])S, w);
        if (inside) { v[0] = v[1] = v[2] = 0; break; }
        int m = 0;
        for (int i = 0; i < 3; ++i) v[i] = 0;
        for (int k = 0; k < n; ++k)
            if (w[k] > 0) {
                for (int i = 0; i < 3; ++i) v[i] += w[k] * S[k][i];
                if (m != k) { memcpy(S[m], S[k], sizeof(S[0])); idx[m] = idx[k]; }
                ++m;
            }
        n = m;
        if (dot3(v, v) < 1e-24) { v[0] = v[1] = v[2] = 0; break; }
    }
    for (int i = 0; i < 3; ++i) p[i] = c[i] + v[i];
    return sqrt(dot3(v, v));
}
double v2p_oracle_hull_closest(const double *V, int nv, const double c[3], double p[3]) { return hull_closest(V, nv, c, p); }

#define NDT (ND + 6) /* generalized velocity with the ball appended: [articulation ND | ball linear 3 | ball angular 3] */
typedef struct {
    int kind;     /* 0 hull vertex x ground, 1 ball x racket cylinder, 2 ball x ground, 3 joint limit (one row: DOF `vert` of joint `body`), 4 ball x the hull of link `body` */
    int body, vert;
    double pos[3]; /* contact point, world */
    double n[3], t1[3], t2[3];
    double gap, mu, rest;
} crow_t;

static int substep_impl(const v2p_omodel *m, const v2p_oparams *p, v2p_ostate *s, const double *pd_target /*[69]*/,
                        const double *ext_force /*[3] world, at root COM*/, const double *ext_torque /*[3] world*/,
                        double *contact_force /*[NB*3] out*/, double *dof_force /*[69] out*/, int *contact_ids /*[NB*4] out: body*64+vertex, -1 padded*/,
                        const v2p_osub_io *io, const v2p_oball_params *bp, v2p_oball *ball, const double *ball_force /*[3] world*/,
                        double *ball_contact /*[9] out: force on the ball from the racket, from the ground, from the humanoid's links*/) {
    double M[ND * ND], C[ND], J[6 * ND];
    double rhs[ND], q[3 * NJ];
    kin_t k;
    const double h = p->h;
    kinematics(m, s, &k);
    mass_and_bias(m, p, s, &k, M, C);

    /* generalized forces */
    for (int i = 0; i < ND; ++i) rhs[i] = -C[i];
    if (ext_force || ext_torque) {
        double d[3], t[3], w6[6] = {0, 0, 0, 0, 0, 0};
        matvec(k.R[0], m->com[0], d);
        if (ext_force) { cross(d, ext_force, t); for (int i = 0; i < 3; ++i) { w6[i] += t[i]; w6[3 + i] += ext_force[i]; } }
        if (ext_torque) for (int i = 0; i < 3; ++i) w6[i] += ext_torque[i];
        body_jacobian(m, &k, 0, J);
        for (int c = 0; c < ND; ++c) for (int l = 0; l < 6; ++l) rhs[c] += J[l * ND + c] * w6[l];
    }
    for (int b = 1; b < NB; ++b) quat2expmap(s->jquat[b - 1], &q[3 * (b - 1)]);
    for (int j = 0; j < 3 * NJ; ++j) {
        double wj = s->vel[6 + j];
        double tar = pd_target ? pd_target[j] : q[j];
        rhs[6 + j] += m->kp[j] * (tar - q[j]) - (m->kd[j] + h * m->kp[j]) * wj;
        M[(6 + j) * ND + 6 + j] += m->armature[j] + h * m->kd[j] + h * h * m->kp[j];
    }
    if (cholesky(M, ND)) return -1;
    chol_solve(M, ND, rhs);
    double v[NDT];
    for (int i = 0; i < ND; ++i) v[i] = s->vel[i] + h * rhs[i];
    for (int i = ND; i < NDT; ++i) v[i] = 0.0;
    if (ball) { /* free flight of the ball: gravity + the aerodynamic force held over this simulate() call */
        for (int i = 0; i < 3; ++i) {
            v[ND + i] = ball->vel[i] + h * ((i == 2 ? p->gravity_z : 0.0) + (ball_force ? ball_force[i] / bp->mass : 0.0));
            v[ND + 3 + i] = ball->angvel[i];
        }
    }

    /* contacts */
    if (contact_force) memset(contact_force, 0, sizeof(double) * NB * 3);
    if (ball_contact) memset(ball_contact, 0, sizeof(double) * 9);
    if (contact_ids) for (int i = 0; i < NB * 4; ++i) contact_ids[i] = -1;
    if (p->enable_contact || p->joint_limits) {
        contacts_t cs;
        cs.n = 0;
        if (p->enable_contact) gen_contacts(m, p, &k, &cs, io);
        /* row list in Gauss-Seidel order: bodies ascending with their hull points in slot order; the ball-racket points right after the
         * hull points of the racket's link; the ball-ground point last */
        /* joint limits (Isaac Gym enforces the MJCF `range` of every DOF; the three hinges of a body are one spherical joint whose DOF
         * positions are the exponential-map components, utils/motion_lib.py:460-488): a DOF whose range is narrower than a full turn
         * carries ONE row against the nearer of its two limits, C = min(q - lo, hi - q), on the joint rate of that DOF (body axes; equal
         * to the rate of the exp-map component to first order), sign +1 (lower) / -1 (upper); bias like a contact's normal row
         * (C/h separated - only an approach that would cross the limit within the substep is stopped -, erp C/h violated); impulse
         * >= 0.  Order: the limit rows of joint b right before the hull points of body b.
         * Activation (speculative, like every other row of this model: contact_offset for hull vertices, the closing distance of a
         * substep for the ball; PhysX: the `contactDistance` of a joint limit): the row exists in this substep only if
         * C < limit_margin + h max(0, approach rate), the rate being the joint rate AFTER the unconstrained update (v*: the implicit
         * PD can change a joint rate by tens of rad/s within a substep).  A violated limit (C < 0) is always active. */
        crow_t rows[NB * MAXC_BODY + 6 + 3 * NJ];
        int nc = 0, ci = 0;
        /* ball x humanoid: every hull (convex hull of a link's contact vertices) the ball can reach within the substep carries one point
         * (PhysX: one per overlapping pair; the MAXH nearest are kept), found at the start of the substep; activation and bias exactly
         * like a racket point; the rows of a link's point are solved right after that link's ground points.  A ball centre INSIDE a hull (more than a radius deep:
         * not reachable through the speculative rows unless it is placed there) is pushed out along the direction from the centre of
         * the hull's body-frame bounding box. */
        g_hull_rows = 0;
#define MAXH 3 /* hull points kept per ball: the nearest links (the engine's LDS block holds three next to the cylinders' two) */
        int nhull = 0, hull_link[MAXH];
        crow_t hull_row[MAXH];
        if (ball && bp->body_contacts && p->enable_contact) {
            for (int b = 0; b < NB; ++b) {
                if (b == bp->racket_link) continue;
                const int v0 = m->hull_offsets[b], nv = m->hull_offsets[b + 1] - v0;
                if (nv <= 0) continue;
                double d[3] = {ball->pos[0] - k.x[b][0], ball->pos[1] - k.x[b][1], ball->pos[2] - k.x[b][2]}, cb[3], pb[3], pt[3], n[3];
                for (int i = 0; i < 3; ++i) cb[i] = k.R[b][i] * d[0] + k.R[b][3 + i] * d[1] + k.R[b][6 + i] * d[2]; /* R^T d */
                double dist = hull_closest(m->hull_verts + 3 * v0, nv, cb, pb);
                if (dist > 1e-6) for (int i = 0; i < 3; ++i) n[i] = (cb[i] - pb[i]) / dist;
                else {
                    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, e[3];
                    for (int q = 0; q < nv; ++q) for (int i = 0; i < 3; ++i) { double x = m->hull_verts[3 * (v0 + q) + i]; lo[i] = fmin(lo[i], x); hi[i] = fmax(hi[i], x); }
                    for (int i = 0; i < 3; ++i) e[i] = cb[i] - 0.5 * (lo[i] + hi[i]);
                    double l = sqrt(dot3(e, e));
                    if (l > 1e-9) for (int i = 0; i < 3; ++i) n[i] = e[i] / l; else { n[0] = 0; n[1] = 0; n[2] = 1; }
                    dist = 0.0;
                    memcpy(pb, cb, sizeof(pb));
                }
                double nw[3], rl[3], wr[3];
                matvec(k.R[b], n, nw);
                matvec(k.R[b], pb, rl);
                for (int i = 0; i < 3; ++i) pt[i] = k.x[b][i] + rl[i];
                cross(k.w[b], rl, wr);
                double vrel = 0;
                for (int i = 0; i < 3; ++i) vrel += (ball->vel[i] - k.xd[b][i] - wr[i]) * nw[i];
                const double gapb = dist - bp->radius;
                if (gapb < p->contact_offset + h * fmax(0.0, -vrel)) {
                    /* one point per overlapping link (PhysX: one per overlapping pair), the MAXH nearest kept, nearest first; ties: lower link */
                    int at = nhull;
                    while (at > 0 && gapb < hull_row[at - 1].gap) --at;
                    if (at < MAXH) {
                        for (int q = (nhull < MAXH ? nhull : MAXH - 1); q > at; --q) { hull_row[q] = hull_row[q - 1]; hull_link[q] = hull_link[q - 1]; }
                        crow_t *hr = &hull_row[at];
                        memset(hr, 0, sizeof(*hr));
                        hr->kind = 4; hr->body = b; hr->vert = 0;
                        memcpy(hr->pos, pt, sizeof(pt)); memcpy(hr->n, nw, sizeof(nw));
                        tangent_basis(nw, hr->t1, hr->t2);
                        hr->gap = gapb; hr->mu = bp->fric_body; hr->rest = bp->rest_body;
                        hull_link[at] = b;
                        if (nhull < MAXH) ++nhull;
                    }
                }
            }
        }
        for (int b = 0; b < NB; ++b) {
            if (p->joint_limits && b >= 1)
                for (int i = 0; i < 3; ++i) {
                    const int j = 3 * (b - 1) + i;
                    const double lo = m->limit_lo[j], hi = m->limit_hi[j];
                    if (hi - lo >= 6.28) continue;
                    const double clo = q[j] - lo, chi = hi - q[j];
                    {
                        const double sg = clo <= chi ? 1.0 : -1.0, gp = clo <= chi ? clo : chi;
                        if (!(gp < p->limit_margin + h * fmax(0.0, -sg * v[6 + j]))) continue;
                    }
                    crow_t *r = &rows[nc++];
                    memset(r, 0, sizeof(*r));
                    r->kind = 3; r->body = b; r->vert = i;
                    r->n[0] = clo <= chi ? 1.0 : -1.0;
                    r->gap = clo <= chi ? clo : chi;
                }
            for (; ci < cs.n && cs.body[ci] == b; ++ci) {
                crow_t *r = &rows[nc++];
                r->kind = 0; r->body = b; r->vert = cs.vert[ci];
                memcpy(r->pos, cs.pos[ci], sizeof(double) * 3);
                r->n[0] = 0; r->n[1] = 0; r->n[2] = 1; r->t1[0] = 1; r->t1[1] = 0; r->t1[2] = 0; r->t2[0] = 0; r->t2[1] = 1; r->t2[2] = 0;
                r->gap = (g_experiment & 1) ? f32_vertex_z(m, s, b, cs.vert[ci]) : cs.pos[ci][2]; r->mu = p->mu; r->rest = 0.0;
                r->gap -= p->rest_offset;
                if (g_experiment & 2) r->gap += g_experiment_param;
            }
            for (int q = 0; q < nhull; ++q) if (hull_link[q] == b) { rows[nc++] = hull_row[q]; ++g_hull_rows; }
            if (ball && b == bp->racket_link)
                for (int j = 0; j < bp->ncyl; ++j) {
                    double cw[3], aw[3], pt[3], n[3];
                    matvec(k.R[b], bp->cyl[j].center, cw);
                    matvec(k.R[b], bp->cyl[j].axis, aw);
                    for (int i = 0; i < 3; ++i) cw[i] += k.x[b][i];
                    double dist = cyl_closest(cw, aw, bp->cyl[j].half_len, bp->cyl[j].radius, ball->pos, pt, n) - bp->radius;
                    /* speculative margin: a 50 m/s ball covers 0.4 m per substep, the head is 4 cm thick.  The point becomes a contact
                     * when the ball can reach it within this substep at the closing speed of the start of the substep; the row's
                     * gap / h bias then lets it approach exactly up to the surface */
                    double rl[3] = {pt[0] - k.x[b][0], pt[1] - k.x[b][1], pt[2] - k.x[b][2]}, wr[3];
                    cross(k.w[b], rl, wr);
                    double vrel = 0;
                    for (int i = 0; i < 3; ++i) vrel += (ball->vel[i] - k.xd[b][i] - wr[i]) * n[i];
                    if (dist < p->contact_offset + h * fmax(0.0, -vrel)) {
                        crow_t *r = &rows[nc++];
                        r->kind = 1; r->body = b; r->vert = j;
                        memcpy(r->pos, pt, sizeof(pt)); memcpy(r->n, n, sizeof(n));
                        tangent_basis(n, r->t1, r->t2);
                        r->gap = dist; r->mu = bp->fric_racket; r->rest = bp->rest_racket;
                    }
                }
        }
        if (ball && ball->pos[2] - bp->radius < p->contact_offset + h * fmax(0.0, -ball->vel[2])) {
            crow_t *r = &rows[nc++];
            r->kind = 2; r->body = -1; r->vert = 0;
            r->pos[0] = ball->pos[0]; r->pos[1] = ball->pos[1]; r->pos[2] = ball->pos[2] - bp->radius;
            r->n[0] = 0; r->n[1] = 0; r->n[2] = 1; r->t1[0] = 1; r->t1[1] = 0; r->t1[2] = 0; r->t2[0] = 0; r->t2[1] = 1; r->t2[2] = 0;
            r->gap = ball->pos[2] - bp->radius; r->mu = bp->fric_ground; r->rest = bp->rest_ground;
        }
        int nrow = nc * 3;
        /* one heap block per substep for the row matrices */
        /* (a grow-only block per thread, cleared per substep: a calloc / free of ~130 KB per substep goes through mmap / munmap, and 128
         * OpenMP threads doing that serialise in the kernel - the batch scaled to 178 env-steps/s per thread instead of ~800) */
        static __thread double *row_block;
        static __thread size_t row_block_cap;
        const size_t row_need = (size_t)(nrow > 0 ? nrow : 1) * (2 * NDT + 3);
        if (row_need > row_block_cap) {
            free(row_block);
            row_block_cap = row_need + row_need / 2;
            row_block = malloc(row_block_cap * sizeof(double));
            if (!row_block) { row_block_cap = 0; return -3; }
        }
        memset(row_block, 0, row_need * sizeof(double));
        double *Jr = row_block;
        double *Tr = Jr + (size_t)(nrow > 0 ? nrow : 1) * NDT;
        double *wii = Tr + (size_t)(nrow > 0 ? nrow : 1) * NDT;
        double *lam = wii + (nrow > 0 ? nrow : 1);
        double *bias = lam + (nrow > 0 ? nrow : 1);
        double gap[NB * MAXC_BODY + 6 + 3 * NJ];
        double rbias[NB * MAXC_BODY + 6 + 3 * NJ]; /* restitution target of a ball row (bias <= rest x approach speed), +inf = none */
        int slot_in_body[NB];
        memset(slot_in_body, 0, sizeof(slot_in_body));
        for (int c = 0; c < nc; ++c) {
            const crow_t *r = &rows[c];
            const double *dirs[3] = {r->n, r->t1, r->t2};
            if (r->kind == 0) {
                if (contact_ids) contact_ids[r->body * 4 + slot_in_body[r->body]] = r->body * 64 + r->vert;
                slot_in_body[r->body]++;
            }
            if (r->body >= 0 && r->kind != 3) body_jacobian(m, &k, r->body, J);
            for (int a = 0; a < 3; ++a) {
                int row = 3 * c + a;
                double *jr = &Jr[row * NDT];
                if (r->kind == 3) { /* rows 1, 2 of a limit stay empty (skipped by the sweep) */
                    if (a > 0) { wii[row] = 1.0; continue; }
                    jr[6 + 3 * (r->body - 1) + r->vert] = r->n[0];
                } else
                /* velocity of the contact point of body B relative to body A along dir, A = ground (kind 0, 2) or the racket's link (kind 1) */
                if (r->kind == 0) { /* B = the link */
                    double rl[3] = {r->pos[0] - k.x[r->body][0], r->pos[1] - k.x[r->body][1], r->pos[2] - k.x[r->body][2]}, rxn[3];
                    cross(rl, dirs[a], rxn); /* point velocity . n = n.xdot + (r x n).w */
                    for (int col = 0; col < ND; ++col) {
                        double sum = 0;
                        for (int l = 0; l < 3; ++l) sum += J[l * ND + col] * rxn[l] + J[(3 + l) * ND + col] * dirs[a][l];
                        jr[col] = sum;
                    }
                } else { /* B = the ball */
                    double rb[3] = {r->pos[0] - ball->pos[0], r->pos[1] - ball->pos[1], r->pos[2] - ball->pos[2]}, rxn[3];
                    if (r->kind == 1 || r->kind == 4) for (int i = 0; i < 3; ++i) rb[i] = -bp->radius * r->n[i]; /* the ball's own contact point */
                    cross(rb, dirs[a], rxn);
                    for (int l = 0; l < 3; ++l) { jr[ND + l] = dirs[a][l]; jr[ND + 3 + l] = rxn[l]; }
                    if (r->kind == 1 || r->kind == 4) { /* minus the point of the link */
                        double rl[3] = {r->pos[0] - k.x[r->body][0], r->pos[1] - k.x[r->body][1], r->pos[2] - k.x[r->body][2]};
                        cross(rl, dirs[a], rxn);
                        for (int col = 0; col < ND; ++col) {
                            double sum = 0;
                            for (int l = 0; l < 3; ++l) sum += J[l * ND + col] * rxn[l] + J[(3 + l) * ND + col] * dirs[a][l];
                            jr[col] = -sum;
                        }
                    }
                }
            }
            if (r->kind == 0 && p->friction_frame == 1) {
                /* friction frame of a hull x ground point from the tangential velocity it has under v* (the rows are linear in their direction) */
                double *j1 = &Jr[(3 * c + 1) * NDT], *j2 = &Jr[(3 * c + 2) * NDT], vx = 0, vy = 0;
                for (int col = 0; col < ND; ++col) { vx += j1[col] * v[col]; vy += j2[col] * v[col]; }
                const double sp = sqrt(vx * vx + vy * vy);
                if (sp > 1e-6) {
                    const double cx = vx / sp, cy = vy / sp;
                    for (int col = 0; col < ND; ++col) {
                        const double a1 = j1[col], a2 = j2[col];
                        j1[col] = cx * a1 + cy * a2;
                        j2[col] = -cy * a1 + cx * a2;
                    }
                    rows[c].t1[0] = cx; rows[c].t1[1] = cy; rows[c].t1[2] = 0;
                    rows[c].t2[0] = -cy; rows[c].t2[1] = cx; rows[c].t2[2] = 0;
                }
            }
            for (int a = 0; a < 3; ++a) {
                int row = 3 * c + a;
                double *jr = &Jr[row * NDT], *tr = &Tr[row * NDT];
                if (r->kind == 3 && a > 0) continue;
                memcpy(tr, jr, sizeof(double) * NDT);
                chol_solve(M, ND, tr);
                if (ball) for (int l = 0; l < 3; ++l) { tr[ND + l] /= bp->mass; tr[ND + 3 + l] /= bp->inertia; }
                double sum = 0;
                for (int col = 0; col < NDT; ++col) sum += jr[col] * tr[col];
                wii[row] = sum;
            }
            double d = r->gap;
            gap[c] = d;
            bias[3 * c] = d >= 0 ? d / h : fmax(p->erp * d / h, -p->max_depen_vel);
            rbias[c] = 1e300;
            if (r->kind == 1 || r->kind == 2 || r->kind == 4) { /* restitution (Newton): an approach faster than the bounce threshold that closes the gap within this substep
                                 * leaves with rest x the approach speed */
                double vn0 = 0;
                for (int col = 0; col < NDT; ++col) vn0 += Jr[3 * c * NDT + col] * v[col];
                if (vn0 < -bp->bounce_threshold && d / h + vn0 < 0) rbias[c] = r->rest * vn0;
                bias[3 * c] = fmin(bias[3 * c], rbias[c]);
            }
        }
        /* solver_type 0, PGS: n_iter sweeps against the biases of the start of the substep.
         * solver_type 1, TGS (temporal Gauss-Seidel, the PhysX solver amass_im.yaml:41 selects; Macklin et al. 2019, "Small steps in
         * physics simulation"), restated with the Jacobians frozen over the substep: the substep is cut into n_iter slices of length
         * hs = h / n_iter, slice k re-evaluates every point's gap with the motion of the previous slices,
         *     d_c(k) = d_c(k-1) + hs * vn_c(after sweep k-1),
         * and runs ONE sweep against  d_c(k) / (h - k hs)  (separated: do not cross the plane in the time that is left)  or
         * max(erp d_c(k) / hs, -max_depenetration)  (penetrating: correct a fraction per slice); impulses accumulate and are clamped
         * on the accumulated value as in PGS.  Every row of the list takes part: hull points, the joint-limit rows (their "gap" is the
         * distance of the DOF to its limit, advanced with the joint rate) and the ball's two-body rows (advanced with the relative normal
         * velocity of the two contact points).  Restitution of a ball row: the bounce decision and its target, rest x the approach speed
         * of v*, are taken ONCE at the start of the substep (as under PGS) and cap the bias of every slice - a ball that has bounced in
         * slice 0 is leaving in slices 1.., its advancing gap would otherwise ask for less than the rebound. */
        int sweep_rev[NB * MAXC_BODY + 6 + 3 * NJ]; /* the points stop by stop, last stop first (see the sweep) */
        {
            int n = 0;
            for (int end = nc; end > 0;) {
                const int stop = rows[end - 1].body;
                int st = end - 1;
                while (st > 0 && rows[st - 1].body == stop) --st;
                for (int c = st; c < end; ++c) sweep_rev[n++] = c;
                end = st;
            }
        }
        for (int it = 0; it < p->n_iter; ++it) {
            if (p->solver_type == 1) {
                double hs = h / p->n_iter;
                for (int c = 0; c < nc; ++c) {
                    if (it > 0) {
                        double vn = 0;
                        for (int col = 0; col < NDT; ++col) vn += Jr[3 * c * NDT + col] * v[col];
                        gap[c] += hs * vn;
                    }
                    bias[3 * c] = gap[c] >= 0 ? gap[c] / (h - it * hs) : fmax(p->erp * gap[c] / hs, -p->max_depen_vel);
                    bias[3 * c] = fmin(bias[3 * c], rbias[c]);
                }
            }
            /* Experiment of round 4 (g_experiment bit 2, value 4; the engine's counterpart is the build switch V2P_LL_ALT_SWEEP): PGS sweeps
             * in ALTERNATING direction over the STOPS (a stop = everything that belongs to one body: the limit rows of its joint, its hull
             * points, the ball point on its hull / the ball x racket points; the ball x ground point is a stop of its own after the last
             * body) - odd sweeps take the stops in descending order, the rows inside a stop in the same order as ever.  A sweep that starts
             * where the previous one ended lets the engine's tree walk go back and forth instead of returning to the first body after
             * every sweep: 8 % fewer instructions.  NOT the model: the body a sweep ends on is solved twice in a row, and after 4 sweeps the
             * distance to the converged solution is 1.5 x that of ascending sweeps in the median standing env (what ~3.6 ascending sweeps
             * reach, for the cost of ~3.7): the saving is paid in full with solver accuracy (dead ends). */
            const int backward = p->solver_type == 0 && (it & 1) && (g_experiment & 4);
            for (int cc = 0; cc < nc; ++cc) {
                const int c = backward ? sweep_rev[cc] : cc;
                for (int a = 0; a < (rows[c].kind == 3 ? 1 : 3); ++a) {
                    int row = 3 * c + a;
                    double rel = bias[row];
                    for (int col = 0; col < NDT; ++col) rel += Jr[row * NDT + col] * v[col];
                    double nl = lam[row] - rel / wii[row];
                    if (io && io->clamp_margin) {
                        const double sw = a == 0 ? fabs(nl) : fabs(fabs(nl) - rows[c].mu * lam[3 * c]);
                        if (sw * wii[row] < *io->clamp_margin) *io->clamp_margin = sw * wii[row];
                    }
                    if (a == 0) { if (nl < 0) nl = 0; }
                    else { double lim = rows[c].mu * lam[3 * c]; if (nl > lim) nl = lim; if (nl < -lim) nl = -lim; }
                    double dl = nl - lam[row];
                    lam[row] = nl;
                    for (int col = 0; col < NDT; ++col) v[col] += Tr[row * NDT + col] * dl;
                }
            }
        }
        for (int c = 0; c < nc; ++c) {
            const crow_t *r = &rows[c];
            double f[3];
            if (r->kind == 3) continue;
            for (int i = 0; i < 3; ++i) f[i] = (lam[3 * c] * r->n[i] + lam[3 * c + 1] * r->t1[i] + lam[3 * c + 2] * r->t2[i]) / h;
            if (r->kind == 0 && contact_force) for (int i = 0; i < 3; ++i) contact_force[3 * r->body + i] += f[i];
            if (r->kind == 1) { /* on the ball +f, on the racket's link -f */
                if (ball_contact) for (int i = 0; i < 3; ++i) ball_contact[i] += f[i];
                if (contact_force) for (int i = 0; i < 3; ++i) contact_force[3 * r->body + i] -= f[i];
            }
            if (r->kind == 2 && ball_contact) for (int i = 0; i < 3; ++i) ball_contact[3 + i] += f[i];
            if (r->kind == 4) {
                if (ball_contact) for (int i = 0; i < 3; ++i) ball_contact[6 + i] += f[i];
                if (contact_force) for (int i = 0; i < 3; ++i) contact_force[3 * r->body + i] -= f[i];
            }
        }
        (void)Jr; /* (kept by the thread for the next substep) */
    }

    /* joint drive torque actually applied (implicit form) */
    if (dof_force)
        for (int j = 0; j < 3 * NJ; ++j) {
            double tar = pd_target ? pd_target[j] : q[j];
            dof_force[j] = m->kp[j] * (tar - q[j] - h * v[6 + j]) - m->kd[j] * v[6 + j];
        }

    /* angular damping + angular velocity clamp, then integrate */
    double sc = 1.0 / (1.0 + h * p->ang_damp);
    for (int i = 3; i < ND; ++i) v[i] *= sc;
    for (int g = 0; g < NB; ++g) {
        double *w = &v[3 + 3 * g];
        double n = sqrt(dot3(w, w));
        if (n > p->max_ang_vel) { double f = p->max_ang_vel / n; w[0] *= f; w[1] *= f; w[2] *= f; }
    }
    memcpy(s->vel, v, sizeof(double) * ND);
    for (int i = 0; i < 3; ++i) s->root_pos[i] += h * v[i];
    double dq[4], rv[3], nq[4];
    for (int i = 0; i < 3; ++i) rv[i] = h * v[3 + i];
    rotvec2quat(rv, dq);
    qmul(dq, s->root_quat, nq); /* world-frame rate: left multiply */
    qnormalize(nq);
    memcpy(s->root_quat, nq, sizeof(nq));
    for (int b = 1; b < NB; ++b) {
        for (int i = 0; i < 3; ++i) rv[i] = h * v[6 + 3 * (b - 1) + i];
        rotvec2quat(rv, dq);
        qmul(s->jquat[b - 1], dq, nq); /* body-frame rate: right multiply */
        qnormalize(nq);
        memcpy(s->jquat[b - 1], nq, sizeof(nq));
    }
    if (ball) {
        double scb = 1.0 / (1.0 + h * bp->ang_damp);
        double *bw = &v[ND + 3];
        for (int i = 0; i < 3; ++i) bw[i] *= scb;
        double nw = sqrt(dot3(bw, bw));
        if (nw > bp->max_ang_vel) for (int i = 0; i < 3; ++i) bw[i] *= bp->max_ang_vel / nw;
        for (int i = 0; i < 3; ++i) { ball->vel[i] = v[ND + i]; ball->angvel[i] = bw[i]; ball->pos[i] += h * v[ND + i]; rv[i] = h * bw[i]; }
        rotvec2quat(rv, dq);
        qmul(dq, ball->quat, nq);
        qnormalize(nq);
        memcpy(ball->quat, nq, sizeof(nq));
    }
    return 0;
}

int v2p_oracle_substep_io(const v2p_omodel *m, const v2p_oparams *p, v2p_ostate *s, const double *pd_target, const double *ext_force,
                          const double *ext_torque, double *contact_force, double *dof_force, int *contact_ids, const v2p_osub_io *io) {
    return substep_impl(m, p, s, pd_target, ext_force, ext_torque, contact_force, dof_force, contact_ids, io, 0, 0, 0, 0);
}

/* aerodynamic force on the ball as humanoid_smpl_im_mvae.py:711-739 evaluates it before every simulate() call (drag with a constant
 * coefficient, Magnus lift whose direction depends on the velocity only and whose size on the spin RATE; vid2player/utils/tennis_ball.py) */
void v2p_oracle_ball_aero(const v2p_oball *ball, double spin_scale, double force[3]) {
    const double PI = 3.14159265358979323846, R = 0.032, rho = 1.21, kf = rho * PI * R * R / 2.0, cd = 0.55;
    double sp = sqrt(dot3(ball->vel, ball->vel));
    double vs = sp == 0.0 ? 1.0 : sp; /* "avoid divide by 0" */
    double vn[3] = {ball->vel[0] / vs, ball->vel[1] / vs, ball->vel[2] / vs};
    const double g[3] = {0, 0, -1};
    double vt[3], lt[3];
    cross(vn, g, vt);
    double vspin = sqrt(dot3(ball->angvel, ball->angvel)) / (2.0 * PI);
    double cl = 1.0 / (2.0 + fabs(vs / (vspin * spin_scale + 1e-6)));
    cl *= vspin > 0 ? -1.0 : 1.0;
    cross(vt, vn, lt);
    for (int i = 0; i < 3; ++i) force[i] = -kf * cd * vs * ball->vel[i] - kf * cl * vs * vs * lt[i];
}

/* one control step with racket + ball: `nsub` substeps in simulate() calls of `sub_per_sim` substeps; the aerodynamic force is
 * re-evaluated at the start of every simulate() call.  ball_per_sim [nsim][13] (pos quat vel angvel after each call),
 * racket_hit_per_sim [nsim] (1 when the racket-ball contact force was non-zero in the call's last substep, as the reference polls the
 * net contact force tensor after each call), ball_contact [9] of the last substep (from the racket, the ground, the humanoid's links). */
int v2p_oracle_step_ball_io(const v2p_omodel *m, const v2p_oparams *p, v2p_ostate *s, const double *pd_target, const double *ext_force,
                         const double *ext_torque, int nsub, int hold, int sub_per_sim, double *contact_force, double *dof_force, int *contact_ids,
                         const v2p_oball_params *bp, v2p_oball *ball, double spin_scale, double *ball_per_sim, int *racket_hit_per_sim,
                         double *ball_contact, int *max_hull_points /*[1] nullable: most ball x hull points active in one substep*/,
                         double *contact_force_sum /*[NB*3] nullable: net contact forces summed over the simulate() calls
                                                                          * (`_contact_forces_sum`, humanoid_smpl_im_mvae.py:781; needs contact_force)*/,
                            const int *forced_ids /*[nsub][NB*4] nullable: hull vertices to use instead of the selection rule (teacher forcing)*/,
                            int *own_ids /*[nsub][NB*4] nullable: the rule's own picks*/, double *margins /*[nsub][NB] nullable*/) {
    double f[3] = {0, 0, 0}, bc[9];
    if (contact_force_sum) memset(contact_force_sum, 0, sizeof(double) * NB * 3);
    if (max_hull_points) *max_hull_points = 0;
    for (int i = 0; i < nsub; ++i) {
        if (i % sub_per_sim == 0) v2p_oracle_ball_aero(ball, spin_scale, f);
        int on = i < hold;
        v2p_osub_io io = {forced_ids ? forced_ids + (size_t)i * NB * 4 : 0, own_ids ? own_ids + (size_t)i * NB * 4 : 0, margins ? margins + (size_t)i * NB : 0, 0};
        int rc = substep_impl(m, p, s, pd_target, on ? ext_force : 0, on ? ext_torque : 0, contact_force, dof_force, contact_ids, &io, bp, ball, f, bc);
        if (rc) return rc;
        if (max_hull_points && g_hull_rows > *max_hull_points) *max_hull_points = g_hull_rows;
        if (i % sub_per_sim == sub_per_sim - 1) {
            int k = i / sub_per_sim;
            if (ball_per_sim) {
                double *o = ball_per_sim + 13 * k;
                memcpy(o, ball->pos, sizeof(double) * 3); memcpy(o + 3, ball->quat, sizeof(double) * 4);
                memcpy(o + 7, ball->vel, sizeof(double) * 3); memcpy(o + 10, ball->angvel, sizeof(double) * 3);
            }
            if (racket_hit_per_sim) racket_hit_per_sim[k] = (bc[0] != 0.0 || bc[1] != 0.0 || bc[2] != 0.0);
            if (contact_force_sum && contact_force) for (int j = 0; j < NB * 3; ++j) contact_force_sum[j] += contact_force[j];
        }
    }
    if (ball_contact) memcpy(ball_contact, bc, sizeof(bc));
    return 0;
}

int v2p_oracle_step_ball(const v2p_omodel *m, const v2p_oparams *p, v2p_ostate *s, const double *pd_target, const double *ext_force,
                         const double *ext_torque, int nsub, int hold, int sub_per_sim, double *contact_force, double *dof_force, int *contact_ids,
                         const v2p_oball_params *bp, v2p_oball *ball, double spin_scale, double *ball_per_sim, int *racket_hit_per_sim,
                         double *ball_contact, int *max_hull_points /*[1] nullable: most ball x hull points active in one substep*/,
                         double *contact_force_sum /*[NB*3] nullable: net contact forces summed over the simulate() calls
                                                                          * (`_contact_forces_sum`, humanoid_smpl_im_mvae.py:781; needs contact_force)*/) {
    return v2p_oracle_step_ball_io(m, p, s, pd_target, ext_force, ext_torque, nsub, hold, sub_per_sim, contact_force, dof_force, contact_ids, bp, ball, spin_scale,
                                   ball_per_sim, racket_hit_per_sim, ball_contact, max_hull_points, contact_force_sum, 0, 0, 0);
}

int v2p_oracle_sizeof_ball_params(void) { return (int)sizeof(v2p_oball_params); }
int v2p_oracle_sizeof_ball(void) { return (int)sizeof(v2p_oball); }

/* ------------------------------------------------------------------ state <-> Isaac-Gym-style tensors */
/* root[13] = pos3 quat4 linvel3 angvel3 (humanoid_smpl.py:66-113), dof_pos = exp-map, dof_vel = joint-frame rate */
void v2p_oracle_set_state(v2p_ostate *s, const double *root13, const double *dof_pos, const double *dof_vel) {
    memcpy(s->root_pos, root13, sizeof(double) * 3);
    memcpy(s->root_quat, root13 + 3, sizeof(double) * 4);
    qnormalize(s->root_quat);
    memcpy(s->vel, root13 + 7, sizeof(double) * 6);
    for (int b = 1; b < NB; ++b) {
        expmap2quat(&dof_pos[3 * (b - 1)], s->jquat[b - 1]);
        for (int i = 0; i < 3; ++i) s->vel[6 + 3 * (b - 1) + i] = dof_vel[3 * (b - 1) + i];
    }
}

void v2p_oracle_get_state(const v2p_omodel *m, const v2p_ostate *s, double *root13, double *dof_pos, double *dof_vel, double *rb_state /*[NB*13]*/) {
    kin_t k;
    kinematics(m, s, &k);
    if (root13) {
        memcpy(root13, s->root_pos, sizeof(double) * 3);
        memcpy(root13 + 3, s->root_quat, sizeof(double) * 4);
        memcpy(root13 + 7, s->vel, sizeof(double) * 6);
    }
    for (int b = 1; b < NB; ++b) {
        if (dof_pos) quat2expmap(s->jquat[b - 1], &dof_pos[3 * (b - 1)]);
        if (dof_vel) for (int i = 0; i < 3; ++i) dof_vel[3 * (b - 1) + i] = s->vel[6 + 3 * (b - 1) + i];
    }
    if (rb_state)
        for (int b = 0; b < NB; ++b) {
            double *o = rb_state + 13 * b;
            memcpy(o, k.x[b], sizeof(double) * 3);
            memcpy(o + 3, k.quat[b], sizeof(double) * 4);
            memcpy(o + 7, k.xd[b], sizeof(double) * 3);
            memcpy(o + 10, k.w[b], sizeof(double) * 3);
        }
}

int v2p_oracle_substep(const v2p_omodel *m, const v2p_oparams *p, v2p_ostate *s, const double *pd_target, const double *ext_force,
                       const double *ext_torque, double *contact_force, double *dof_force, int *contact_ids) {
    return v2p_oracle_substep_io(m, p, s, pd_target, ext_force, ext_torque, contact_force, dof_force, contact_ids, 0);
}

/* one control step = nsub substeps; the residual wrench is held for the first `hold` substeps
 * (Isaac Gym consumes applied forces in the next simulate() only: SURVEY.md section 7 hard parts).
 * forced_ids [nsub][NB*4] (nullable): contact vertices to use in each substep instead of the selection rule;
 * own_ids [nsub][NB*4], margins [nsub][NB] (nullable): what the rule selects in the state of each substep and how narrowly. */
int v2p_oracle_step_io(const v2p_omodel *m, const v2p_oparams *p, v2p_ostate *s, const double *pd_target, const double *ext_force,
                       const double *ext_torque, int nsub, int hold, double *contact_force, double *dof_force, int *contact_ids,
                       const int *forced_ids, int *own_ids, double *margins, double *clamp_margin /*[1] nullable: min over the substeps*/) {
    if (clamp_margin) *clamp_margin = 1e30;
    for (int i = 0; i < nsub; ++i) {
        int on = i < hold;
        v2p_osub_io io = {forced_ids ? forced_ids + (size_t)i * NB * 4 : 0, own_ids ? own_ids + (size_t)i * NB * 4 : 0, margins ? margins + (size_t)i * NB : 0,
                          clamp_margin};
        int rc = v2p_oracle_substep_io(m, p, s, pd_target, on ? ext_force : 0, on ? ext_torque : 0, contact_force, dof_force, contact_ids, &io);
        if (rc) return rc;
    }
    return 0;
}

int v2p_oracle_step(const v2p_omodel *m, const v2p_oparams *p, v2p_ostate *s, const double *pd_target, const double *ext_force,
                    const double *ext_torque, int nsub, int hold, double *contact_force, double *dof_force, int *contact_ids) {
    return v2p_oracle_step_io(m, p, s, pd_target, ext_force, ext_torque, nsub, hold, contact_force, dof_force, contact_ids, 0, 0, 0, 0);
}

/* A batch of independent humanoids, OpenMP over envs (the cpu_baseline leg of bench.py and the larger parity tests): env e uses
 * models[model_of ? model_of[e] : 0]; every per-env array is contiguous [n][...]; the optional arrays as in v2p_oracle_step_io.
 * rb_state [n][NB*13] (nullable) receives the rigid-body state after the step.  Returns the number of failed envs. */
int v2p_oracle_step_batch(const v2p_omodel *const *models, const int *model_of, const v2p_oparams *p, v2p_ostate *states, int n,
                          const double *pd_target /*[n][69]*/, const double *ext_force /*[n][3]*/, const double *ext_torque /*[n][3]*/,
                          int nsub, int hold, double *contact_force /*[n][NB*3]*/, double *dof_force /*[n][69]*/, int *contact_ids /*[n][NB*4]*/,
                          const int *forced_ids /*[n][nsub][NB*4]*/, int *own_ids, double *margins /*[n][nsub][NB]*/,
                          double *root13 /*[n][13]*/, double *dof_pos /*[n][69]*/, double *dof_vel /*[n][69]*/, double *rb_state /*[n][NB*13]*/,
                          int num_threads, double *clamp_margin /*[n] nullable*/) {
    int failed = 0;
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : failed)
    for (int e = 0; e < n; ++e) {
        const v2p_omodel *m = models[model_of ? model_of[e] : 0];
        size_t E = (size_t)e;
        int rc = v2p_oracle_step_io(m, p, &states[e], pd_target ? pd_target + E * 3 * NJ : 0, ext_force ? ext_force + E * 3 : 0,
                                    ext_torque ? ext_torque + E * 3 : 0, nsub, hold, contact_force ? contact_force + E * NB * 3 : 0,
                                    dof_force ? dof_force + E * 3 * NJ : 0, contact_ids ? contact_ids + E * NB * 4 : 0,
                                    forced_ids ? forced_ids + E * nsub * NB * 4 : 0, own_ids ? own_ids + E * nsub * NB * 4 : 0,
                                    margins ? margins + E * nsub * NB : 0, clamp_margin ? clamp_margin + E : 0);
        if (rc) { ++failed; continue; }
        if (root13 || dof_pos || dof_vel || rb_state)
            v2p_oracle_get_state(m, &states[e], root13 ? root13 + E * 13 : 0, dof_pos ? dof_pos + E * 3 * NJ : 0, dof_vel ? dof_vel + E * 3 * NJ : 0,
                                 rb_state ? rb_state + E * NB * 13 : 0);
    }
    return failed;
}

int v2p_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ diagnostics for invariant tests */
/* kinetic energy, potential energy, linear momentum[3], angular momentum about the world origin[3] */
void v2p_oracle_diagnostics(const v2p_omodel *m, const v2p_oparams *p, const v2p_ostate *s, double *out8) {
    kin_t k;
    kinematics(m, s, &k);
    double ke = 0, pe = 0, P[3] = {0, 0, 0}, L[3] = {0, 0, 0};
    for (int b = 0; b < NB; ++b) {
        double I6[36], d[3], Ic[9], wd[3], vc[3], c[3], Iw[3], t[3];
        spatial_inertia(m, &k, b, I6, d, Ic);
        cross(k.w[b], d, wd);
        for (int i = 0; i < 3; ++i) { vc[i] = k.xd[b][i] + wd[i]; c[i] = k.x[b][i] + d[i]; }
        matvec(Ic, k.w[b], Iw);
        ke += 0.5 * m->mass[b] * dot3(vc, vc) + 0.5 * dot3(k.w[b], Iw);
        pe += -m->mass[b] * p->gravity_z * c[2];
        cross(c, vc, t);
        for (int i = 0; i < 3; ++i) { P[i] += m->mass[b] * vc[i]; L[i] += m->mass[b] * t[i] + Iw[i]; }
    }
    out8[0] = ke; out8[1] = pe;
    for (int i = 0; i < 3; ++i) { out8[2 + i] = P[i]; out8[5 + i] = L[i]; }
}

int v2p_oracle_sizeof_model(void) { return (int)sizeof(v2p_omodel); }
int v2p_oracle_sizeof_state(void) { return (int)sizeof(v2p_ostate); }
int v2p_oracle_sizeof_params(void) { return (int)sizeof(v2p_oparams); }
