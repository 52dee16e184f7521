/*
 * usim_oracle.h -- C interface of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is a plain-C, scalar, one-environment-at-a-time restatement of the reference's
 * Ultrasound env.step()/reset() hot path.  It exists so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the HIP path
 * against an independently written implementation.  Nothing under robotic-ultrasound-imaging_amd/
 * may include, link or dlopen it.
 *
 * PARITY STATUS: "parity unpinned" at the MuJoCo boundary.  The reference's arithmetic lives in
 * un-vendored dependencies (MuJoCo 2.0 binary, mujoco-py, robosuite fork) that can
 * be neither compiled nor imported here, and the reference ships no tests or golden trajectories.
 * What IS pinned: the env-level formulas that live in /root/reference/src (reward, observation
 * layout, bookkeeping, termination, quaternion helpers, trajectory/reset sampling) and the decoded
 * checkpoint data in tests/golden/reference_pins.npz (reset observations, force-depth line).
 *
 * Built twice from the same source: -DREAL=double (libusim_oracle_f64.so, the checker) and
 * -DREAL=float (libusim_oracle_f32.so, used to separate precision effects from logic errors).
 */
#ifndef USIM_ORACLE_H
#define USIM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define USO_OBS_DIM 19
#ifndef USO_MAXC
#define USO_MAXC 8          /* contact slots per env */
#endif
#ifndef USO_MAXCAND
#define USO_MAXCAND 16      /* penetrating elements considered before the USO_MAXC deepest are kept */
#endif
#define USO_NSCALAR 40      /* scalar state words per env exported by uso_get_state */

/* impedance_mode of the OSC controller (rl_config.yaml:41, main.py:33, utils/plot.py:203-211,303-313) */
enum { USO_MODE_TRACKING = 0, USO_MODE_FIXED = 1, USO_MODE_VARIABLE_Z = 2, USO_MODE_WRENCH = 3 };
/* torso model: 0 = rigid/absent (BASELINE config #2), 1 = 99 top-face elements dynamic (config #3) */
enum { USO_TORSO_NONE = 0, USO_TORSO_TOP = 1, USO_TORSO_FULL = 2 };   /* 2: all 270 shell elements on the free torso body, element-table contacts (round 4) */
enum { USO_ROBOT_PANDA = 0, USO_ROBOT_UR5E = 1 };

typedef struct uso_config {
    int32_t mode;                 /* USO_MODE_* */
    int32_t torso;                /* USO_TORSO_* */
    int32_t horizon;              /* rl_config.yaml:27 (1000) */
    int32_t early_termination;    /* rl_config.yaml:52 */
    int32_t deterministic_trajectory;        /* rl_config.yaml:54, ultrasound.py:762-764 */
    int32_t torso_solref_randomization;      /* rl_config.yaml:55, ultrasound.py:291-297 */
    int32_t initial_probe_pos_randomization; /* rl_config.yaml:56, ultrasound.py:870-887 */
    int32_t friction_randomization;          /* BASELINE config #5 (new knob) */
    int32_t torso_drop;           /* 0 (default since round 4): the torso base stays at its spawn height -- it stands on the caps of its tilted rim capsules (torso_dz);
                                   * 1: free fall over the 4.7 mm spawn gap of ultrasound.py:313, then rest (rounds 1-3: a flat bottom); 2: at rest 4.7 mm lower from the start */
    int32_t pgs_iters;            /* iterations of the contact solver (default 24 of cone_solver 2; cone_solver 1: sweeps; cone_solver 0: full sweeps interleaved with normal-only ones, N N F F N F F at 4) */
    int32_t ik_iters;             /* fixed reset-IK iteration count */
    int32_t env_offset;           /* global index of env 0 (multi-GPU shards) */
    int32_t torso_shape;          /* 0 box (soft_box.xml, use_box_torso True), 1 cylinder (soft_human_torso.xml) */
    int32_t robot;                /* USO_ROBOT_*: the two robots ultrasound.py:137 admits */
    uint64_t seed;                /* rl_config.yaml:1 */
    double control_dt;            /* 1/control_freq = 0.002 (rl_config.yaml:26) */
    double kp_fixed;              /* rl_config.yaml:38 / main.py:31 */
    double damping_ratio;         /* rl_config.yaml:39 */
    double kp_min, kp_max;        /* kp_limits rl_config.yaml:42 */
    double out_max_pos, out_max_ori;   /* output_max rl_config.yaml:36 */
    double stiffness, damping;    /* soft_box.xml:9 solrefsmooth (1324.17, 17.59) */
    double elem_friction;         /* soft_box.xml:10 (0.01) */
    double probe_friction;        /* ultrasound_probe_gripper.xml:8 (1e-4) */
    double probe_radius, probe_halflen;   /* stand-in for the missing probe mesh (.MISSING_LARGE_BLOBS:1): tip radius, half-length */
    double probe_radius2, probe_height;   /* ... radius of the upper edge of the flared blade and its height above the tip axis */
    int32_t substeps;             /* physics steps per env.step(): int(control_timestep / model_timestep) of robosuite MujocoEnv.step [RESTATED, SURVEY C.1];
                                   * control_dt is the CONTROL timestep (ultrasound.py:542), the physics step is control_dt / substeps (0 or 1: one) */
    int32_t lattice_ramp;         /* STUDY switch, oracle only (tests/studies/lattice_ramp_study.py): 1 = evaluate MuJoCo's impedance ramp d(r) of solimp (0.9 0.95 0.001 0.5 2)
                                   * on every lattice row (the lattice matrix is then assembled and factorised per step); 0 = the product's model, d fixed at
                                   * d_max = 0.95 so that the inverse is a constant */
    double study_fix_tc;          /* STUDY switch, oracle only (tests/studies/sustained_load_study.py): time constant of the joint-equality ("fix") rows of the lattice; 0 = MuJoCo's
                                   * default solref time constant 0.02 s, which the product uses */
    double probe_friction2;       /* sliding friction of the probe's SECOND colliding geom, see probe_geoms (MuJoCo default 1.0) */
    int32_t probe_geoms;          /* 1: one probe geom collides; 2: two coincident ones (ultrasound_probe_gripper.xml:8-9: `probe_collision` AND `probe_visual` -- the
                                   * latter carries no contype / conaffinity = 0, so with MuJoCo's defaults it collides as well, with the default friction 1.0): every
                                   * element pair then has two contacts of the same geometry.  Restated as one contact: normal row with half the regulariser (two equal
                                   * rows in parallel), friction rows of the high-friction contact alone (the other one's cone, mu = 0.01, is saturated at once),
                                   * cone limit (mu_1 + mu_2) / 2 of the TOTAL normal force */
    int32_t cone_solver;          /* contact solver: 2 (default since round 5) block Jacobi with an exact line search -- every contact solves its own cone block from the current
                                   * residual, the step along the joint direction is the quadratic's exact minimiser capped at 1; what the kernels run.  1: exact-cone block
                                   * Gauss-Seidel -- per visit a ray update then the friction QCQP, pgs_iters sweeps (round 4; the converged reference of the tests at 30 sweeps).
                                   * Both rest at the optimum of MuJoCo's convex problem.  0: the schedule of rounds 1-3 (row relaxations + radial scaling of the friction, N N F F N F F),
                                   * kept for A/B studies -- it rests at a different point */
    double probe_halfwidth;       /* round 4: half-width of the flat part of the probe's face across the blade (the face is a 2 probe_halflen x 2 probe_halfwidth rectangle
                                   * with edges of radius probe_radius; 0 = the blade of round 3) */
    int32_t pair_model;           /* probe_geoms = 2: 1 (default since round 5) = the two coincident contacts of a probe-element pair as two contacts of the convex problem, as in MuJoCo
                                   * (the environment's friction word is then the first contact's); 0 = the merged contact of rounds 3-4 */
    int32_t warm_start;           /* STUDY switch, oracle only: 1 = the contact solver starts from the forces of the previous physics step (matched by element) */
    double study_stop_eps;        /* STUDY switch, oracle only: > 0 = the Jacobi solve of an environment stops when the predicted decrease of the dual cost falls below it (pgs_iters stays the cap) */
    double probe_tip;             /* round 4: the probe's lowest point lies this far beyond grip_site along the site's z axis (0: the tip is the site) */
    double armature_scale;        /* round 5: rotor inertia armature_i = armature_scale * 5 / (i + 1) kg m^2 on arm joint i, added to the diagonal of the mass matrix (MuJoCo joint armature).
                                   * [RECALLED: robosuite >= 1.2 RobotModel.__init__ -- the reference imports robosuite.utils.observables, a 1.2 API -- sets armature 5 / (i + 1), frictionloss
                                   * 0.1 and damping 0.1 on robot joints that do not specify them; the snapshot does not vendor robosuite.]  Default 1; 0 = none (rounds 1-4).  Evidence: all
                                   * three shipped checkpoints replay closer to their MuJoCo statistics with it */
    double joint_frictionloss;    /* round 5: dry friction of every arm joint, N m (MuJoCo joint frictionloss; restated joint by joint: usim_oracle.c joint_friction).  Default 0.1; 0 = none */
} uso_config;

void  uso_default_config(uso_config* c);
void* uso_create(const uso_config* c, int n_envs);
void  uso_destroy(void* h);
int   uso_action_dim(void* h);
int   uso_num_elements(void* h);           /* dynamic torso elements (0 or 99) */

/* All array I/O below is double precision regardless of REAL; the f32 build casts on entry/exit. */

/* reset envs where mask[i]!=0 (mask NULL = all).  Draws come from the counter-based stream keyed
 * (seed, env_offset+i, episode index).  obs_out (n x 19, may be NULL) gets the reset observation. */
int uso_reset(void* h, const uint8_t* mask, double* obs_out);

/* reset with explicit per-env draws instead of the RNG stream, params[i] =
 * {start xyz, end xyz, u0, noise xyz, stiffness, damping, mu} (13 doubles, world coordinates). */
int uso_reset_explicit(void* h, const uint8_t* mask, const double* params, double* obs_out);

/* one env.step() for every env.  act: n x A.  Outputs (any may be NULL):
 *   obs n x 19 (already the reset observation where done, SB3 VecEnv semantics),
 *   rew n, done n, term_obs n x 19 (pre-reset observation; valid where done),
 *   contacts n x (1+USO_MAXC) int32: count then ascending shell-element ids of probe<->element pairs.
 * auto_reset!=0 applies the VecEnv auto-reset; 0 leaves finished envs frozen for inspection. */
int uso_step(void* h, const double* act, double* obs, double* rew, uint8_t* done,
             double* term_obs, int32_t* contacts, int auto_reset);

/* scalar state n x USO_NSCALAR (layout in usim_oracle.c: uso_get_state) + lattice n x E x 2 (s, sdot) */
int uso_get_state(void* h, double* scalars, double* lattice);
int uso_set_state(void* h, const double* scalars, const double* lattice);

/* counter-based synthetic actions for rollout step `step`: n x A uniform in the
 * action box, keyed (seed, env_offset+i, step). */
int uso_random_actions(void* h, int64_t step, double* act);

/* diagnostics for unit tests: per-env forward quantities at the current state with zero torque
 * out[0..2] eef pos (world), [3..11] eef rotmat row-major, [12..60] M 7x7, [61..67] bias,
 * [68..109] J 6x7 (site, world axes), [110..112] contact force on probe, [113..115] ee torque sensor */
int uso_debug_forward(void* h, int env, double* out);
/* diagnostics: signed probe distance of the 99 elements (element order) at the current state, the contact list the forward
 * pass keeps (count + element indices, not shell ids); returns the overflow flag */
int uso_debug_contacts(void* h, int env, const double* act, double* out /* [USO_MAXC][8] */);
int uso_element_distances(void* h, int env, double* dist_out, int32_t* contacts_out);
/* study hooks (tests/cone_qp.py, tests/studies/pair_lab.py, full_torso_lab.py): the dual contact problem of the forward pass at the current state of `env` under the
 * action `act` -- the top-face model's (layout at g_dual_dump in usim_oracle.c; returns the number of pairs) and the full torso's (layout at g_full_dump; `cap` doubles
 * available; returns the number of virtual contacts, 0 if the buffer is too small).  One global pointer each: not thread-safe. */
int uso_debug_dual(void* h, int env, const double* act, double* out);
int uso_debug_full(void* h, int env, const double* act, double* out, long cap);
/* signed distance of a point (site frame) from the probe stand-in and the direction the collision uses there (unit vector) */
double uso_probe_sdf(void* h, const double* p_site, double* grad_out);

#ifdef __cplusplus
}
#endif
#endif
