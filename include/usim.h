/*
 * usim.h -- C ABI of libusim, the MI355X-native batched Ultrasound simulator.
 *
 * This is the drop-in boundary of SURVEY.md section 8(b): one handle simulates n independent copies of the
 * reference's `Ultrasound` robosuite environment (src/my_environments/ultrasound.py:29) and one call
 * advances all of them by one env.step().  The reference has no FFI of its own (it is pure Python on top of
 * mujoco-py); each entry point below names the reference interface it replaces.  The Python host class
 * robotic-ultrasound-imaging_amd/vec_env.py binds exactly these symbols with ctypes and exposes the
 * stable-baselines3 VecEnv surface used by src/rl.py:130-143.
 *
 * Conventions
 *   - plain C, no exceptions across the boundary: every function returns 0 (USIM_OK) or a negative
 *     usim_status; usim_strerror() maps it to text.  Per-environment numerical faults never fail a call;
 *     they are reported in the status word of the environment (usim_get_state, field `status`).
 *   - all `*_dev` pointers are device (HBM) pointers owned by the caller (PyTorch-ROCm tensors);
 *     all other pointers are host pointers.  The library owns all simulator state and allocates nothing
 *     inside usim_step().
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); work is only
 *     enqueued, the call does not synchronise.  A handle is not thread-safe.
 *   - observations are float32 [n][19] row-major, actions float32 [n][A] row-major (A = 6, or 7 for
 *     variable_z), rewards float32 [n], dones uint8 [n].
 */
#ifndef USIM_H
#define USIM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define USIM_OBS_DIM 19   /* ultrasound.py:394-401: 3+3+3+1+1+1+7 */
#define USIM_MAXC 8       /* contact slots per environment */
#define USIM_NSCALAR 40   /* scalar state words per environment in usim_get_state/usim_set_state */
#define USIM_RESET_PARAMS 13
#define USIM_LOG_WIDTH 53

typedef enum usim_status {
    USIM_OK = 0,
    USIM_ERR_INVALID = -1,      /* bad argument / configuration */
    USIM_ERR_NO_DEVICE = -2,    /* no HIP device, or device index out of range */
    USIM_ERR_HIP = -3,          /* a HIP runtime call failed (see usim_last_hip_error) */
    USIM_ERR_ALLOC = -4,
    USIM_ERR_UNSUPPORTED = -5
} usim_status;

/* impedance_mode of the OSC_POSE controller: rl_config.yaml:41 ("tracking"), main.py:33 ("fixed"),
 * utils/plot.py:208-211,303-313 ("variable_z"), utils/plot.py:267-268 ("wrench": the action, +-10, replaces desired_force/desired_torque of the OSC law) */
enum { USIM_MODE_TRACKING = 0, USIM_MODE_FIXED = 1, USIM_MODE_VARIABLE_Z = 2, USIM_MODE_WRENCH = 3 };
/* torso model: BASELINE.json configs[1] (rigid, contact solver off) / configs[2] (soft torso) */
/* USIM_TORSO_FULL: all 270 shell elements on the free torso body of ultrasound.py:426-431, element-table contacts (soft_box.xml:9, ultrasound.py:300-314) -- one wave
 * per environment, an order of magnitude slower than the top-face model that the metric is quoted on; Panda, substeps 1, one step per launch */
enum { USIM_TORSO_NONE = 0, USIM_TORSO_TOP = 1, USIM_TORSO_FULL = 2 };
/* robots (ultrasound.py:137) */
enum { USIM_ROBOT_PANDA = 0, USIM_ROBOT_UR5E = 1 };

/* Mirrors the `robosuite:` block of src/rl_config.yaml:18-57 (the kwargs of Ultrasound.__init__,
 * ultrasound.py:99-136) plus the knobs the MJCF assets fix in the reference. */
typedef struct usim_config {
    int32_t struct_size;                       /* sizeof(usim_config) of the header the CALLER was built against.  The caller sets it before
                                                * usim_default_config(); that call and usim_create() return USIM_ERR_INVALID -- without writing
                                                * anything -- when it differs from the library's, so a binding built against another layout of
                                                * this struct fails loudly instead of passing shifted fields or being overrun */
    int32_t mode;                              /* controller_configs.impedance_mode */
    int32_t torso;                             /* USIM_TORSO_* */
    int32_t horizon;                           /* rl_config.yaml:27 */
    int32_t early_termination;                 /* rl_config.yaml:52 */
    int32_t deterministic_trajectory;          /* rl_config.yaml:54 */
    int32_t torso_solref_randomization;        /* rl_config.yaml:55 */
    int32_t initial_probe_pos_randomization;   /* rl_config.yaml:56 */
    int32_t friction_randomization;            /* BASELINE.json configs[4] */
    int32_t torso_drop;                        /* 0 (default since round 4): the torso base stays at its spawn height -- ultrasound.py:313 spawns the nominal bottom plane 4.7 mm
                                                * above the table, but the caps of the tilted rim capsules of the bottom face reach 4.9 - 5.8 mm below that plane and carry
                                                * the torso from the first step; 1: free fall over the 4.7 mm, then rest (rounds 1-3); 2: at rest 4.7 mm
                                                * lower from the start */
    int32_t pgs_iters;                         /* iterations of the contact solver per forward pass (default 24): block Jacobi with a line search on the dual of MuJoCo's
                                                * convex contact problem -- every contact solves its own 3 x 3 cone block at the same time (ray update, then the friction QCQP with
                                                * the normal fixed), the step along the joint direction is the minimiser of the quadratic with the slope taken block by block, capped at 1.  24 iterations rest
                                                * 2e-3 N (99th percentile) from the optimum, which is what MuJoCo's Newton solver converges to.
                                                * USIM_TORSO_FULL: sweeps of a block Gauss-Seidel over the probe and the element-table contacts, started from the forces of the
                                                * previous physics step (94 % of the environments follow a converged solve's decisions at 24) */
    int32_t ik_iters;                          /* reset inverse-kinematics iterations */
    int32_t env_offset;                        /* global index of env 0 of this handle (multi-GPU shard) */
    int32_t lanes_per_env;                     /* kernel mapping: 0 automatic; 16 lanes per environment (arm mathematics distributed over the group); 64 (soft torso: the split
                                                * kernel with 8-lane groups, 32 environments per workgroup; automatic beyond 4096 envs); 32 (soft torso: the
                                                * same, arm side and lattice / contact side in two waves that share a SIMD; the automatic choice for the soft torso);
                                                * 8 (soft torso; arm mathematics replicated per lane) or 1 (rigid torso: one environment per lane) */
    int32_t torso_shape;                       /* use_box_torso (rl_config.yaml:57): 0 box (soft_box.xml), 1 cylinder (soft_human_torso.xml) */
    int32_t waves_per_simd;                    /* 16-lane step kernel: register budget for 1 or 2 waves per SIMD; 0 auto (1 up to 4096 envs, 2 beyond) */
    int32_t robot;                             /* USIM_ROBOT_*: robots of ultrasound.py:137 */
    uint64_t seed;                             /* rl_config.yaml:1 */
    double control_dt;                         /* 1 / control_freq (rl_config.yaml:26) */
    double kp_fixed, damping_ratio;            /* rl_config.yaml:38-39 */
    double kp_min, kp_max;                     /* rl_config.yaml:42 */
    double out_max_pos, out_max_ori;           /* rl_config.yaml:36 */
    double stiffness, damping;                 /* soft_box.xml:9 solrefsmooth */
    double elem_friction, probe_friction;      /* soft_box.xml:10, ultrasound_probe_gripper.xml:8 */
    double probe_radius, probe_halflen;        /* stand-in for the missing probe mesh (ultrasound_probe_gripper.xml:3,8), a flared blade = convex hull of two
                                                * parallel capsules along the site x axis: tip capsule radius / half-length (its axis one radius above grip_site) ... */
    double probe_radius2, probe_height;        /* ... upper capsule: radius, height of its axis above the tip capsule's (probe_height > |probe_radius2 - probe_radius|) */
    int32_t substeps;                          /* physics steps per env.step(): int(control_timestep / model_timestep) of robosuite MujocoEnv.step, model timestep 2 ms
                                                * (1 with the shipped control_freq 500, rl_config.yaml:26; 25 with the env default 20, ultrasound.py:119).  control_dt above is
                                                * the CONTROL timestep (ultrasound.py:542); the physics step is control_dt / substeps.  substeps > 1: 16-lane kernels only
                                                * (USIM_ERR_UNSUPPORTED otherwise) */
    int32_t probe_geoms;                       /* colliding geoms of the probe body.  2 (default): ultrasound_probe_gripper.xml:8-9 declares `probe_collision` AND `probe_visual`
                                                * on the same mesh, and the visual one carries no contype = conaffinity = 0 -- with MuJoCo's defaults it collides too, with the
                                                * default friction (1, 0.005, 0.0001): every probe-element pair has two coincident contacts.  With pair_model = 0 restated as one contact whose normal
                                                * row has half the regulariser (two equal rows in parallel), whose friction rows are those of the high-friction contact (the other
                                                * cone, mu = 0.01, saturates at once) and whose cone limit is (mu_1 + mu_2) / 2 of the total normal force.  1: a single probe geom */
    double probe_friction2;                    /* sliding friction of the second geom (MuJoCo default 1.0) */
    double probe_halfwidth;                    /* round 4: half-width of the flat part of the probe's face ACROSS the blade -- the face is a flat 2 probe_halflen x 2 probe_halfwidth
                                                * rectangle whose edges have the radius probe_radius (0: the blade of round 3, a tip capsule) */
    double probe_tip;                          /* round 4: the lowest point of the probe lies this far beyond grip_site along the site's z axis (0: the tip is the site) */
    int32_t pair_model;                        /* probe_geoms = 2 only.  1 (default since round 5): the two coincident contacts of a probe-element pair are TWO contacts of the
                                                * convex problem, as in MuJoCo -- the same three rows twice, each with a single contact's regulariser, cones
                                                * mu_A = max(probe_friction, elem_friction) (the per-environment friction word) and mu_B = max(probe_friction2, elem_friction).
                                                * 0: the merged contact of rounds 3-4 (half the normal regulariser, cone (mu_A + mu_B) / 2) */
    int32_t reserved0;                         /* (keeps the struct a multiple of 8 bytes) */
    double armature_scale;                     /* rotor inertia armature_i = armature_scale * 5 / (i + 1) kg m^2 on arm joint i (MuJoCo joint armature: added to the diagonal of the
                                                * mass matrix -- the plant's and, through sim.data.qM, the OSC controller's).  [RECALLED: robosuite >= 1.2 RobotModel.__init__ sets armature
                                                * 5 / (i + 1), frictionloss 0.1 and damping 0.1 on robot joints that do not specify them; the reference imports robosuite.utils.observables
                                                * (ultrasound.py:18), a 1.2 API; robosuite is not vendored in the snapshot.]  Default 1 since 0.5; 0 = none (earlier versions).  All three shipped
                                                * checkpoints replay closer to their MuJoCo statistics with it */
    double joint_frictionloss;                 /* dry friction of every arm joint, N m (MuJoCo joint frictionloss, 0.1 by the same robosuite default): one bounded constraint row per joint in
                                                * MuJoCo, restated joint by joint.  Default 0.1; 0 = none (earlier versions) */
} usim_config;

typedef struct usim_handle usim_handle;

/* Per-step I/O block.  Required: obs, rew, done.  `act` may be NULL only for usim_rollout_random /
 * usim_time_steps (actions are then drawn in-kernel).  Optional outputs may be NULL. */
typedef struct usim_step_io {
    const float* act_dev;      /* [n][A]   replaces the `action` argument of env.step (robosuite MujocoEnv.step) */
    float* obs_dev;            /* [n][19]  GymWrapper-flattened observation; the RESET observation where done */
    float* rew_dev;            /* [n]      Ultrasound.reward, ultrasound.py:230-269 */
    uint8_t* done_dev;         /* [n]      ultrasound.py:549-550 + horizon */
    float* term_obs_dev;       /* [n][19]  SB3 infos[i]["terminal_observation"]; written where done */
    int32_t* contacts_dev;     /* [n][1+USIM_MAXC] count, then ascending shell-element ids (ultrasound.py:673-736) */
    float* ep_return_dev;      /* [n]      SB3 Monitor infos[i]["episode"]["r"]; written where done */
    int32_t* ep_length_dev;    /* [n]      SB3 Monitor infos[i]["episode"]["l"]; written where done */
    float* act_out_dev;        /* [n][A]   usim_rollout_random only: the actions drawn in-kernel (completes the transition) */
    int32_t* status_dev;       /* [n]      per-environment status word after this step (bits stay set until the episode ends): bit 0 = more than USIM_MAXC simultaneous probe contacts, bit 1 =
                                *          (USIM_TORSO_FULL only) more element-table contacts than the kernel keeps (120), bit 2 = numerical fault (non-finite / run-away state of the
                                *          arm or -- USIM_TORSO_FULL -- of the free torso body or a slider; the episode is ended and, with auto_reset, restarted) */
    float* log_dev;            /* [n][USIM_LOG_WIDTH] per-step episode record, the channels of the reference's save_data CSV dump
                                *          (ultrasound.py:552-614): ee_pos3 goal_pos3 ee_vel3 goal_vel vbar ee_quat4 goal_quat4 quat_dist Fz goal_Fz
                                *          Fz_mean dFz goal_dFz is_contact q7 torques7 time% pos/ori/vel/force/dforce reward action7 */
} usim_step_io;

/* fills *c with the shipped configuration (src/rl_config.yaml).  c->struct_size must hold sizeof(usim_config) on entry:
 *     usim_config c = { sizeof c };  usim_default_config(&c); */
int usim_default_config(usim_config* c);

/* Replaces suite.make("Ultrasound", **options) + GymWrapper (src/rl.py:36-40) for n environments on HIP
 * device `device`.  Environments start un-reset; call usim_reset first.  Every handle owns its model tables (lattice inverse,
 * element geometry) in device memory: any number of handles, with the same or different configurations (torso_shape, torso, mode),
 * may be alive on one GPU at the same time.  The upload is complete when usim_create returns. */
int usim_create(const usim_config* cfg, int n_envs, int device, usim_handle** out);
void usim_destroy(usim_handle* h);

/* Switch the kernel mapping of a live soft-torso handle between lanes_per_env 32 (split kernel) and 16 (waves_per_simd 0 / 1 / 2 as in
 * usim_config).  The mappings compute the same bits, so a rollout may change between them at any step -- e.g. the two-waves-per-SIMD
 * 16-lane build while a collective's workgroups are resident, the split kernel otherwise (bench.py, N > 1).  Takes effect with the next
 * call that enqueues work. */
int usim_set_mapping(usim_handle* h, int lanes_per_env, int waves_per_simd);

int usim_num_envs(const usim_handle* h);
int usim_action_dim(const usim_handle* h);     /* GymWrapper.action_space.shape[0] */
int usim_num_elements(const usim_handle* h);   /* dynamic torso elements per env (0, 99 or 270) */

/* Replaces env.reset() (ultrasound.py:416-478) for the envs where mask_dev[i] != 0 (NULL = all).
 * obs_dev [n][19] (may be NULL) receives the reset observation of the selected envs. */
int usim_reset(usim_handle* h, const uint8_t* mask_dev, float* obs_dev, void* stream);

/* Same, with explicit per-env draws instead of the seeded stream: params_dev [n][13] =
 * start xyz, end xyz (world), u0, position noise xyz, stiffness, damping, contact friction. */
int usim_reset_explicit(usim_handle* h, const uint8_t* mask_dev, const float* params_dev, float* obs_dev, void* stream);

/* Replaces SubprocVecEnv.step_async + step_wait over n envs (src/rl.py:130): one env.step() each, with
 * SB3 auto-reset semantics when auto_reset != 0. */
int usim_step(usim_handle* h, const usim_step_io* io, int auto_reset, void* stream);

/* Synthetic workload of BASELINE.md section 4: fills act_dev [n][A] with i.i.d. uniform actions from the
 * counter-based stream keyed (seed, global env id, step). */
int usim_random_actions(usim_handle* h, int64_t step, float* act_dev, void* stream);

/* Enqueues nsteps consecutive steps whose actions are drawn in-kernel from the same stream as
 * usim_random_actions(first_step + k) (io->act_dev ignored).  block_advance == 0: every step writes the same
 * buffers.  block_advance != 0: the non-NULL buffers of io are rollout blocks [nsteps][n][...] and step k writes
 * slice k (SB3 RolloutBuffer layout, the unit that is all-gathered across GPUs). */
int usim_rollout_random(usim_handle* h, int64_t first_step, int nsteps, const usim_step_io* io, int block_advance, void* stream);

/* Refill the reset bank now (the launch usim_step issues by itself every 256 steps: the initial states of the episodes that will reuse the ring slots
 * consumed since the last refill) and restart the 256-step period.  For callers that record a FIXED sequence of steps once and replay it -- a HIP
 * graph captured around T x usim_step (policy.GraphedCollector): the period counter lives on the host and does not advance at replay, so such a
 * sequence starts and ends with this call (every ring is then valid at every replay, whatever T).  Capture-safe: on a stream that is being captured
 * the library records kernel launches only (no events, no synchronising call).  Nothing in the reference corresponds to it (a reset there
 * recompiles the model, ultrasound.py:416-478). */
int usim_refill_bank(usim_handle* h, void* stream);

/* usim_rollout_random enqueues its steps in launches of up to `steps` consecutive steps each (1 .. 256, default 256 = the refill period of the reset
 * bank, which a launch never crosses; the 16- and 32-lane mappings -- the others always launch step by step): inside a launch the lattice tables stay in LDS and launch
 * latency is paid once, every step still writes its state and its slice of the rollout block to HBM.  A launch lasts as long as its slowest workgroup -- the one whose
 * environments had the most contacts --, so long launches also average that out: 4096 envs, us per step: 32 steps per launch 15.6, 64 14.7, 128 14.1, 256 13.6.  The
 * results do not depend on the value (bit for bit). */
int usim_set_steps_per_launch(usim_handle* h, int steps);
/* the value in force (the default, a usim_set_steps_per_launch call or the USIM_STEPS_PER_LAUNCH environment variable read by usim_create); < 0: error code */
int usim_get_steps_per_launch(const usim_handle* h);

/* Device time spent so far in the reset-bank refill launches of this handle (one every 256 steps with auto-reset: the initial-pose IK and
 * zero-torque forward pass of the episodes that will start next), from HIP events around them on their stream.  Blocks
 * until the refill launches issued so far have finished.  bench.py subtracts it from the event time of its step blocks to get the duration of
 * the step kernel alone -- the number a rocprofv3 kernel trace reports. */
int usim_refill_time(usim_handle* h, double* total_ms, long long* launches);

/* Like usim_rollout_random, bracketed by HIP events on `stream`; blocks until done and returns the elapsed
 * device time in milliseconds (bench.py roofline leg). */
int usim_time_steps(usim_handle* h, int64_t first_step, int nsteps, const usim_step_io* io, int block_advance, void* stream, float* elapsed_ms);

/* Checkpoint / inspection.  Host buffers; synchronises the device.
 * scalars [n][USIM_NSCALAR]: q[0..6] qd[7..13] q0[14..20] traj_start[21..23] traj_end[24..26] u0[27] vbar[28]
 * fzbar[29] fzprev[30] dfz[31] stiffness[32] damping[33] mu[34] t[35] has_touched[36] episode[37]
 * ep_return[38] status[39];  lattice [n][E][2] = (s, sdot) per element (may be NULL when E == 0). */
int usim_get_state(usim_handle* h, float* scalars, float* lattice);
int usim_set_state(usim_handle* h, const float* scalars, const float* lattice);
/* USIM_TORSO_FULL: host buffers [n][USIM_FULL_BODY_WORDS] of float64 -- the free torso body (MuJoCo's free joint, ultrasound.py:426-431): position (world), quaternion
 * (w x y z), linear velocity (world axes), angular velocity (body frame) = 13 words; then the warm start of the contact solve (the contact forces of the previous physics step:
 * [270][4] every element's table-contact force and friction multiplier, [8] the probe slots' elements, [16][4] their forces by geom).  float64: the device holds the position
 * relative to the robot base in float32, and base + position is exact in float64 -- what usim_get_body_state hands out restores the same bits.  usim_get_state /
 * usim_set_state carry the 270 sliders in `lattice`. */
#define USIM_FULL_BODY_WORDS (13 + 4 * 270 + 8 + 64)
int usim_get_body_state(usim_handle* h, double* body);
int usim_set_body_state(usim_handle* h, const double* body);

/* Diagnostics: runs one step (in-kernel synthetic actions of `step`, auto-reset on) on the default stream, blocks, and returns
 * shader-clock stamps taken in workgroup 0: ticks[0..16] by wave 0 at the phase boundaries of the single-wave step
 * kernels, ticks[20..29] / ticks[30..38] by the arm wave / the lattice wave of the split kernel at their barriers (min(max_ticks, 64) words
 * are written, max_ticks >= 17).  Only the profiling build of the library (make -C csrc prof) carries the stamps; the production build
 * returns USIM_ERR_UNSUPPORTED. */
int usim_profile_step(usim_handle* h, const usim_step_io* io, int64_t step, uint64_t* ticks, int max_ticks);

/* ---- the caller's side of env.step() on the device: SB3 VecNormalize + MlpPolicy forward + sampling + rollout-buffer
 * writes + GAE, fused into a few kernels per rollout step (csrc/usim_policy.hip).  Everything is a device pointer into the CALLER's tensors
 * (torch parameters, policy.DeviceVecNormalize statistics, policy.DeviceRolloutBuffer slices); the library keeps no copy. ---- */
typedef struct usim_policy_net {               /* stable_baselines3 ActorCriticPolicy("MlpPolicy", net_arch pi=[256,128] vf=[256,128], tanh) of the shipped
                                                * checkpoints (src/rl.py:143; trained_rl_models/.zip policy.pth); torch.nn.Linear layout weight[out][in] */
    const float *pi_w1, *pi_b1, *pi_w2, *pi_b2;        /* mlp_extractor.policy_net: [256][19], [256], [128][256], [128] */
    const float *act_w, *act_b;                        /* action_net: [A][128], [A] */
    const float *vf_w1, *vf_b1, *vf_w2, *vf_b2;        /* mlp_extractor.value_net */
    const float *val_w, *val_b;                        /* value_net: [1][128], [1] */
    const float* log_std;                              /* [A] */
    const float* w2_packed;                            /* REQUIRED: USIM_POLICY_PACKED floats filled by usim_policy_pack from pi_w2 / vf_w2 as they are NOW -- the two layer-2
                                                        * matrices in the order the matrix-core operands are consumed, every word split into two float16 (hi + lo: layer 2 runs
                                                        * as three float16 products with float32 accumulation, 2^-22 relative).  Pack again after every change of the weights
                                                        * (policy.FusedRollout: before every eager launch, and as the first policy node of a recorded rollout) */
} usim_policy_net;
#define USIM_POLICY_PACKED (2 * 128 * 256)
int usim_policy_pack(const usim_policy_net* net, float* packed_dev, void* stream);

typedef struct usim_norm_stats {               /* stable_baselines3 VecNormalize (src/rl.py:140,177-184): RunningMeanStd of observations and of discounted returns */
    double *obs_mean, *obs_var, *obs_count;            /* [19], [19], scalar -- updated in place when training */
    double *ret_mean, *ret_var, *ret_count;            /* scalars */
    double* returns;                                   /* [n] discounted return accumulators */
    double* scratch;                                   /* USIM_POLICY_SCRATCH doubles of workspace, zero-initialised by the caller once (required when training) */
    double clip_obs, clip_reward, gamma, epsilon;      /* 10, 10, 0.99, 1e-8 in the shipped pickles */
} usim_norm_stats;

#define USIM_POLICY_SCRATCH 1280               /* 32 x 19 x 2 partial sums + an arrival counter */

typedef struct usim_policy_out {
    float* act_env_dev;        /* [n][A] REQUIRED: the sampled action clipped to the action box = the `action` argument of the following usim_step */
    float* nobs_dev;           /* [n][19] normalised observation            -> RolloutBuffer.observations[t]   (may be NULL, like the rest) */
    float* act_dev;            /* [n][A]  unclipped sample                  -> RolloutBuffer.actions[t] */
    float* value_dev;          /* [n]     value_net output                  -> RolloutBuffer.values[t] */
    float* logp_dev;           /* [n]     log-probability of the sample     -> RolloutBuffer.log_probs[t] */
    float* episode_start_dev;  /* [n]     1.0 where the previous step ended an episode -> RolloutBuffer.episode_starts[t] */
} usim_policy_out;

/* VecNormalize.normalize_obs (+ RunningMeanStd.update when training) -> ActorCriticPolicy.forward -> sample -> clip, for the n environments whose raw
 * observations are obs_dev [n][19].  prev_done_dev [n]: done flags of the previous step (NULL: every environment starts an episode).  The Gaussian
 * noise comes from a counter-based stream keyed (seed, env_offset + env, counter + *counter_base_dev): a new value per call -- counter from the host,
 * counter_base_dev (may be NULL) a device word that a recorded sequence of calls (HIP graph) advances between replays.  deterministic != 0: the
 * mean action.  training: 0 = statistics frozen, 1 = update them with obs_dev first, 2 = they already contain obs_dev (see usim_policy_reward). */
int usim_policy_step(const usim_policy_net* net, const usim_norm_stats* st, const float* obs_dev, const uint8_t* prev_done_dev, int n, int act_dim,
                     const float* act_low_dev, const float* act_high_dev, uint64_t seed, uint32_t counter, const uint32_t* counter_base_dev, int env_offset,
                     int training, int deterministic, const usim_policy_out* out, void* stream);
/* The same step with BOTH halves of VecNormalize.step_wait inside the policy launch (two launches per rollout step: this one and usim_step), statistics
 * updated in place.  update_obs: RunningMeanStd.update(obs_dev) before it is normalised (the observation an env step returned; 0 where the statistics have
 * already seen obs_dev).  have_prev: the reward side of the step BEFORE this observation -- rew_prev_dev / done_prev_dev [n] (the env's buffers, not yet
 * overwritten), nrew_prev_dev [n] receives the normalised reward; raw_sum_dev (may be NULL) += the sum of the raw rewards.  The workgroups exchange their
 * partial sums inside the launch and wait for one another (bounded), which needs all of them resident: n <= USIM_POLICY_FUSED_MAX_ENVS, else
 * USIM_ERR_UNSUPPORTED (use usim_policy_step + usim_policy_reward), and nothing else running on the device beside the launch.  work_dev: USIM_POLICY_FUSED_WORK(n)
 * doubles, zero-initialised once; counter + *counter_base_dev must not repeat for a workspace (it stamps the arrival flags).  After a synchronisation,
 * the 32-bit word behind the 2 x USIM_POLICY_FUSED_ROWS(n) flags that follow the rows is non-zero if a wait ever ran out. */
#define USIM_POLICY_FUSED_MAX_ENVS 8192
#define USIM_POLICY_FUSED_ROWS(n) (((n) + 31) / 32)
#define USIM_POLICY_FUSED_WORK(n) (USIM_POLICY_FUSED_ROWS(n) * 49 + 2)
typedef struct usim_policy_fused {
    double* work_dev;                /* USIM_POLICY_FUSED_WORK(n) doubles: one row of 48 per 32 environments (partial sums), then the arrival flags and a status word */
    const float* rew_prev_dev; const uint8_t* done_prev_dev; float* nrew_prev_dev; double* raw_sum_dev;
    int32_t update_obs, have_prev, norm_reward, reserved_;
} usim_policy_fused;
int usim_policy_step_fused(const usim_policy_net* net, const usim_norm_stats* st, const usim_policy_fused* f, const float* obs_dev, const uint8_t* prev_done_dev, int n,
                           int act_dim, const float* act_low_dev, const float* act_high_dev, uint64_t seed, uint32_t counter, const uint32_t* counter_base_dev,
                           int env_offset, int deterministic, const usim_policy_out* out, void* stream);
/* VecNormalize's reward side after the env step: returns = gamma returns + rew, RunningMeanStd.update(returns), nrew = clip(rew / sqrt(ret_var + eps)),
 * returns reset where done.  raw_sum_dev (may be NULL): += sum of the raw rewards.  next_obs_dev (may be NULL; used when training): the observation the step
 * returned -- its RunningMeanStd.update is made in the same launch (as VecNormalize.step_wait does, before the observation is normalised); the
 * usim_policy_step that consumes that observation is then called with training = 2 ("statistics already updated with this observation"). */
int usim_policy_reward(const usim_norm_stats* st, const float* rew_dev, const uint8_t* done_dev, int n, int training, int norm_reward, float* nrew_dev,
                       double* raw_sum_dev, const float* next_obs_dev, void* stream);
/* RolloutBuffer.compute_returns_and_advantage over [T][n] buffers (GAE(lambda), SB3's recursion) */
int usim_policy_gae(const float* rewards_dev, const float* values_dev, const float* episode_starts_dev, const float* last_values_dev, const uint8_t* last_done_dev,
                    int T, int n, float gamma, float gae_lambda, float* advantages_dev, float* returns_dev, void* stream);

const char* usim_strerror(int status);
const char* usim_last_hip_error(const usim_handle* h);
const char* usim_version(void);

#ifdef __cplusplus
}
#endif
#endif /* USIM_H */
