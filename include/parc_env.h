/*
 * parc_env.h — C-ABI of libparc_env.so: the MI355X (gfx950) motion-tracking environment.
 *
 * This is the drop-in boundary for the reference's per-env step.  The reference has no FFI: the
 * hot path sits behind the Python class contract BaseEnv (PARC/motion_tracker/envs/base_env.py:20-72)
 * as driven by BaseAgent (PARC/motion_tracker/learning/base_agent.py:291-370).  A thin Python shim
 * (parc_amd/envs/hip_parkour_env.py, ctypes) implements that class contract on top of the entry points
 * below; each entry point cites the reference code it replaces (file:line relative to /root/reference).
 *
 * Conventions
 *   - plain C, no torch types: pointers + sizes.  "_host" pointers are copied during the call and may
 *     be freed afterwards; "_dev" pointers are device memory OWNED BY THE CALLER (PyTorch tensors) that
 *     must stay alive while bound; the library never frees them.
 *   - every launch goes to the hipStream_t passed in (void* here so that the header needs no HIP
 *     include); nothing in step/reset synchronises with the host.
 *   - return 0 on success, negative ParcStatus on error; parc_last_error() gives a thread-local text.
 *   - single caller thread per handle; handles are independent (one process per GPU needs no locks).
 *   - quaternions are (x, y, z, w); all floating point is fp32; row-major [N][...] arrays.
 */
#ifndef PARC_ENV_H
#define PARC_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PARC_ABI_VERSION 6
#define PARC_MAX_BODIES 16   /* 15 quats + root position share one 16-lane group */
#define PARC_MAX_DOFS 40     /* dof velocities live in floats [88,128) of a 128-float frame record */
#define PARC_MAX_TAR_STEPS 6 /* 2 + steps skeletons <= 8 lane groups of 8 */
#define PARC_MAX_KEY_BODIES 8
#define PARC_MAX_FK_PATHS 8
#define PARC_MAX_FK_DEPTH 8
#define PARC_MAX_GEOMS 24

typedef enum {
    PARC_OK = 0,
    PARC_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
    PARC_ERR_HIP = -2,         /* a HIP runtime call failed */
    PARC_ERR_STATE = -3,       /* call order (e.g. step before bind/load) */
    PARC_ERR_NO_DEVICE = -4    /* no gfx950 device visible */
} ParcStatus;

typedef enum { PARC_JOINT_ROOT = 0, PARC_JOINT_HINGE = 1, PARC_JOINT_SPHERICAL = 2, PARC_JOINT_FIXED = 3 } ParcJointType;
typedef enum { PARC_DONE_NULL = 0, PARC_DONE_FAIL = 1, PARC_DONE_SUCC = 2, PARC_DONE_TIME = 3 } ParcDoneFlag; /* base_env.py:14-18 */
typedef enum { PARC_LOOP_CLAMP = 0, PARC_LOOP_WRAP = 1 } ParcLoopMode;                                        /* motion_lib.py:14-16 */
typedef enum { PARC_GEOM_BOX = 0, PARC_GEOM_SPHERE = 1, PARC_GEOM_CAPSULE = 2 } ParcGeomType;

typedef struct ParcEnv ParcEnv;

/* Character tables — what KinCharModel.load_char_file / init produce (kin_char_model.py:154-190,235). */
typedef struct {
    int32_t num_bodies;                                  /* B, body 0 = root */
    int32_t dof_size;                                    /* D */
    int32_t parent[PARC_MAX_BODIES];
    float local_translation[PARC_MAX_BODIES][3];
    float local_rotation[PARC_MAX_BODIES][4];
    int32_t joint_type[PARC_MAX_BODIES];                 /* ParcJointType */
    float joint_axis[PARC_MAX_BODIES][3];
    int32_t dof_idx[PARC_MAX_BODIES];
    int32_t fk_paths[PARC_MAX_FK_PATHS][PARC_MAX_FK_DEPTH]; /* root-to-leaf body chains, -1 padded */
} ParcCharModel;

/* Rigid-body / actuator parameters the reference hands to Isaac Gym through the MJCF
 * (ig_char_env.py:100-143, humanoid.xml) and the sim block of dm_env_default.yaml:175-186. */
/* ControlMode of the reference (ig_char_env.py:21-26).  pd: the action is the PD position target (exp-map per spherical joint), clipped to the action
 * bounds (implicit drive, PhysX DOF_MODE_POS).  vel: velocity target, drive force = damping * (action - dof_vel) (DOF_MODE_VEL, stiffness 0).  torque: the
 * clipped action is the joint torque (DOF_MODE_EFFORT).  pd_exp / pd_1d: the explicit PD torque of ig_char_env.py:399-421, computed once per control
 * step from the state at its start, target = the action as given (not clipped, :500-503), limited to the motor efforts; pd_1d needs a character
 * whose joints all have one dof (the reference asserts it, :246-250). */
#define PARC_CTRL_PD 0
#define PARC_CTRL_VEL 1
#define PARC_CTRL_TORQUE 2
#define PARC_CTRL_PD_EXP 3
#define PARC_CTRL_PD_1D 4
typedef struct {
    int32_t num_geoms;
    int32_t geom_body[PARC_MAX_GEOMS];
    int32_t geom_type[PARC_MAX_GEOMS];
    float geom_pos[PARC_MAX_GEOMS][3];   /* sphere/box centre, capsule end 0 (body frame) */
    float geom_pos2[PARC_MAX_GEOMS][3];  /* capsule end 1 */
    float geom_size[PARC_MAX_GEOMS][3];  /* sphere r | box half extents | capsule r */
    float geom_density[PARC_MAX_GEOMS];
    float dof_stiffness[PARC_MAX_DOFS];  /* PD kp (MJCF joint stiffness) */
    float dof_damping[PARC_MAX_DOFS];    /* PD kd */
    float dof_armature[PARC_MAX_DOFS];
    float dof_effort[PARC_MAX_DOFS];     /* motor gear = max torque */
    float dof_lower[PARC_MAX_DOFS];
    float dof_upper[PARC_MAX_DOFS];
    float gravity_z;                     /* ig_env.py:142-145 */
    float sim_dt;                        /* 1 / sim_freq */
    int32_t sim_steps;                   /* sim_freq / control_freq (ig_env.py:113) */
    int32_t substeps;                    /* dm_env_default.yaml:186 */
    int32_t solver_iterations;           /* num_position_iterations */
    float friction;                      /* ig_util.py:10 */
    float restitution;
    float contact_offset;
    float max_depenetration_velocity;
    float angular_damping;               /* ig_char_env.py:142 */
    float max_angular_velocity;          /* ig_char_env.py:143 */
    int32_t control_mode;                /* env.control_mode (ig_char_env.py:21-26,95): PARC_CTRL_*; what `action` means (ig_char_env.py:488-506) */
} ParcDynamicsParams;

/* Environment constants — the keys IGParkourEnv.__init__ reads (ig_parkour_env.py:43-149). */
typedef struct {
    uint32_t abi_version;       /* PARC_ABI_VERSION */
    uint32_t struct_size;       /* sizeof(ParcEnvConfig) */
    int32_t device;             /* HIP device ordinal */
    int32_t num_envs;           /* N local envs of this handle */
    ParcCharModel model;
    int32_t num_key_bodies;
    int32_t key_body_ids[PARC_MAX_KEY_BODIES];
    int32_t num_tar_obs_steps;
    int32_t tar_obs_steps[PARC_MAX_TAR_STEPS];
    int32_t num_rays;
    const float *ray_points_host;            /* [R][2] geom_util.get_xy_points_cone:251 */
    double control_dt;                       /* 1 / control_freq (python double, cast like torch does) */
    float episode_length;
    float min_obs_h, max_obs_h;
    float pose_w, vel_w, root_pos_w, root_vel_w, key_pos_w;   /* already divided by their sum (:91-101) */
    float joint_err_w[PARC_MAX_BODIES];      /* [J] */
    float dof_err_w[PARC_MAX_DOFS];          /* [D] */
    float contact_weights[PARC_MAX_BODIES];  /* [B] */
    float pose_termination_dist[PARC_MAX_BODIES]; /* [J] */
    float root_pos_termination_dist, root_rot_termination_angle;
    int32_t enable_early_termination, pose_termination, track_root, track_root_h;
    int32_t report_tracking_error;
    float fail_rate_ema_weight;              /* dm_env.py:88 */
    float min_motion_weight;                 /* dm_env.py:26 */
    float rand_root_pos_offset_scale;        /* mgdm_dm_util.py:102-106 */
    int32_t rand_reset;                      /* dm_env.py:500-504 */
    int32_t demo_mode;                       /* dm_env.py:479-480 */
    const float *env_offsets_host;           /* [N][3] ig_parkour_env.py:389-398 (global env ids of this shard) */
    float action_low[PARC_MAX_DOFS], action_high[PARC_MAX_DOFS]; /* ig_char_env.py:307-347 */
    int32_t body_pos_from_fk;                /* 1: rigid-body positions := FK(char state) inside the step
                                                (reduced-coordinate sim / kinematic mode); 0: read bound buffer */
    int32_t enable_dynamics;                 /* 0 = kinematic-only step (state injected by the caller) */
    ParcDynamicsParams dynamics;             /* used when enable_dynamics */
    uint64_t seed;
    /* `contact_bodies` (ig_parkour_env.py:62, dm_env_default.yaml:8; [] by default): bit b set = body b may touch the ground.  Non-zero
     * switches on the fall rule of compute_done (mgdm_dm_util.py:349-360): FAIL when some OTHER body carries a contact-force component
     * above 0.1 and some other body is lower than termination_height above the terrain under it (RefCharEnv.update_done :147-152). */
    uint32_t contact_body_mask;
    float termination_height;                /* ig_parkour_env.py:63 */
    /* `global_obs` (ig_parkour_env.py:83; false by default): compute_char_obs (ig_char_env.py:586-589, :603) and compute_tar_obs
     * (mgdm_dm_util.py:417) leave root rotation, root velocities, root / key offsets in the global frame. */
    int32_t global_obs;
    /* `global_root_height_obs` (ig_parkour_env.py:84, passed on as compute_char_obs's root_height_obs, :904; false by default): the root
     * height is one more observation in FRONT of the character block (ig_char_env.py:620-622): every later offset moves by one. */
    int32_t global_root_height_obs;
    /* `use_contact_info` (ig_parkour_env.py:72-73; true by default): false drops the target / character contact blocks of the observation
     * (:927-946) and the contact term of the reward (:1032-1040).  `enable_tar_obs` (:83; true by default): false drops the look-ahead
     * target block (mgdm_dm_util.py:482-493) and, with it, the target contact block (:928-930). */
    int32_t use_contact_info, enable_tar_obs;
    /* Developer / test switches, "key=value;key=value" (NULL or "" = none; the product never sets any).  They are part of the configuration
     * so that a measurement can state them (parc_env_describe): the library reads no environment variable.
     *   segments=none|capsules   drop the collision segments (all / the sole edges of boxes)      dtang=<float>   tangential contact damping
     *   kernel=coop|thread       force one of the general dynamics kernels                         man_period=<n>  contact discovery every n substeps
     *   no_residual=1            no root-position residual (precision test)                        ema_leader=1 | curriculum_two_launches=1
     *                                                                                              force one of the curriculum launch paths */
    const char *dev_options;
} ParcEnvConfig;

/* Motion clips as MotionLib._load_motion_file receives them (motion_lib.py:255-401); the library
 * derives root_vel / root_ang_vel / dof_vel on the device and packs 512-byte frame records. */
typedef struct {
    int32_t num_motions;
    const int32_t *num_frames_host;   /* [M] */
    const int32_t *fps_host;          /* [M] */
    const int32_t *loop_modes_host;   /* [M] ParcLoopMode */
    const double *weights_host;       /* [M] un-normalised */
    const float *root_pos_host;       /* [F][3], clips concatenated */
    const float *root_rot_host;       /* [F][4] */
    const float *joint_rot_host;      /* [F][J][4] */
    const float *contacts_host;       /* [F][B] or NULL (=> zeros, motion_lib.py:345-347) */
} ParcMotionClips;

/* Device buffers owned by the caller.  NULL = that optional output is not written. */
typedef struct {
    /* simulator-side character state (ig_char_env.py:173-196): read by step, written by reset/dynamics */
    float *char_root_pos, *char_root_rot, *char_root_vel, *char_root_ang_vel; /* [N][3|4|3|3] */
    float *char_dof_pos, *char_dof_vel;                                       /* [N][D] */
    float *char_body_pos;                                                     /* [N][B][3] */
    float *contact_forces;                                                    /* [N][B][3] */
    /* bookkeeping (dm_env.py:76-78, ig_parkour_env.py:616-621) */
    int32_t *motion_ids, *terrain_ids;                                        /* [N] */
    float *time_offsets;                                                      /* [N] */
    int32_t *timestep;                                                        /* [N] */
    float *time;                                                              /* [N] optional */
    int64_t *ep_num;                                                          /* [N] optional */
    /* outputs of the step (BaseEnv.step contract) */
    float *obs;                                                               /* [N][obs_dim] */
    float *reward;                                                            /* [N] */
    int32_t *done;                                                            /* [N] ParcDoneFlag */
    float *reward_terms;                                                      /* [7][N] pose,vel,root_pos,root_vel,key_pos,contact,total */
    float *tracking_error;                                                    /* [N][7] optional */
    /* optional mirrors of the reference's ref_* tensors (ig_parkour_env.py:560-571) */
    float *ref_root_pos, *ref_root_rot, *ref_root_vel, *ref_root_ang_vel;
    float *ref_joint_rot, *ref_dof_pos, *ref_dof_vel, *ref_body_pos, *ref_contacts;
    float *ray_hfs;                                                           /* [N][R] optional */
} ParcEnvBuffers;

const char *parc_last_error(void);
int parc_abi_version(void);

/* IGParkourEnv.__init__ (ig_parkour_env.py:43-149) minus Isaac Gym */
int parc_env_create(const ParcEnvConfig *cfg, ParcEnv **out);
void parc_env_destroy(ParcEnv *env);
/* obs width for this configuration (IGEnv.get_obs_space, ig_env.py:86-96) */
int parc_env_obs_dim(const ParcEnv *env);

/* MotionLib._load_motion_file (motion_lib.py:255-401) */
int parc_env_load_motions(ParcEnv *env, const ParcMotionClips *clips);
/* DeepMimicEnv.build_terrain_square / load_terrain result (dm_env.py:157-316,447-463):
 * hf [X][Y] x-major, motion_offsets [M][T][2] */
int parc_env_load_terrain(ParcEnv *env, const float *hf_host, int32_t X, int32_t Y, float min_x, float min_y,
                          float dx, float dy, const float *motion_offsets_host, int32_t M, int32_t T);
/* IGCharEnv._build_sim_tensors / IGParkourEnv._build_data_buffers tensor views */
int parc_env_bind_buffers(ParcEnv *env, const ParcEnvBuffers *bufs);
/* The state view (SURVEY 8(b) "get_state / set_state"): the character, reference and bookkeeping state lives in the
 * caller's device buffers bound above; this returns those pointers, reading them is get_state, writing them before
 * parc_env_step is set_state (that is how the kinematic configuration injects the character state). */
int parc_env_get_buffers(ParcEnv *env, ParcEnvBuffers *out);

/* IGEnv.step (ig_env.py:66-84): clip action -> [dynamics] -> _post_physics_step.  action_dev [N][D]
 * may be NULL when enable_dynamics == 0. */
int parc_env_step(ParcEnv *env, const float *action_dev, void *stream);
/* IGParkourEnv.reset (ig_parkour_env.py:809-829) with device RNG.  env_ids_dev int64 [k]; k < 0 = all. */
int parc_env_reset(ParcEnv *env, const int64_t *env_ids_dev, int32_t k, void *stream);
/* Same with the random draws injected (parity tests): all arrays device, length k. */
int parc_env_reset_with(ParcEnv *env, const int64_t *env_ids_dev, int32_t k, const int32_t *motion_ids_dev,
                        const int32_t *terrain_ids_dev, const float *t0_dev, const float *xy_noise_dev, void *stream);
/* BaseAgent._reset_done_envs (base_agent.py:366-370) fused on the device: resets every env whose done flag was
 * raised by the last parc_env_step, using the step's own compacted done list (no nonzero(), no host sync). */
int parc_env_reset_done(ParcEnv *env, void *stream);
/* _update_observations(env_ids) (ig_env.py:396-403): rays + obs rows only. k < 0 = all. */
int parc_env_compute_obs(ParcEnv *env, const int64_t *env_ids_dev, int32_t k, void *stream);

/* dm_env.py:84-88 curriculum state */
int parc_env_get_fail_rates(ParcEnv *env, float *out_host, int32_t M);
int parc_env_set_fail_rates(ParcEnv *env, const float *in_host, int32_t M);
/* per-motion tables derived at load (motion_lib.py:361-384); any pointer may be NULL */
int parc_env_get_motion_info(ParcEnv *env, float *lengths_host, float *weights_host, int32_t M);
/* `never_done` (ig_parkour_env.py:59,980): done flags read NULL after update_done, so the agent resets nothing and
 * parc_env_reset_done resets nobody; the fail-rate curriculum still sees the episode ends, as in the reference. */
int parc_env_set_never_done(ParcEnv *env, int32_t never_done);
/* reset sampling switches of DeepMimicEnv: `rand_reset` / `demo_mode` (dm_env.py:28-29,479-505; set_demo_mode :737-741) and the scale of the random
 * root offset; the per-env start time as a fraction of the clip length used when rand_reset is off (set_motion_start_time_fraction, dm_env.py:743-744,
 * read at :502; the record mode's retry schedule sets it) */
int parc_env_set_rand_reset(ParcEnv *env, int32_t rand_reset, int32_t demo_mode, float root_pos_offset_scale);
int parc_env_set_start_time_fraction(ParcEnv *env, const float *frac_dev /* [N] or NULL */);

/* Stand-alone operators on device arrays (cfg-1 plumbing and the KinCharModel / MotionLib mirrors):
 * kin_char_model.py:586,601,617; motion_lib.py:94 */
int parc_dof_to_rot(ParcEnv *env, const float *dof_dev, float *joint_rot_dev, int32_t n, void *stream);
int parc_rot_to_dof(ParcEnv *env, const float *joint_rot_dev, float *dof_dev, int32_t n, void *stream);
int parc_forward_kinematics(ParcEnv *env, const float *root_pos_dev, const float *root_rot_dev,
                            const float *joint_rot_dev, float *body_pos_dev, float *body_rot_dev, int32_t n, void *stream);
int parc_calc_motion_frame(ParcEnv *env, const int32_t *motion_ids_dev, const float *times_dev, int32_t n,
                           float *root_pos_dev, float *root_rot_dev, float *root_vel_dev, float *root_ang_vel_dev,
                           float *joint_rot_dev, float *dof_vel_dev, float *contacts_dev, void *stream);
/* TEST ENTRY POINT: one of the quaternion device functions of parc_math.hpp (the restatements of torch_util.py:6-530 that
 * the kernels inline) applied element-wise to device arrays, so that the reference's edge-case vectors (w < 0, tiny angles,
 * |sin| < 1e-3, |cos| >= 1, exp maps beyond pi; tests/golden/quat_ops.npz) reach the DEVICE code and not only the CPU oracle.
 * a: [n][4] (ops on quaternions) or [n][3] (PARC_QOP_NORMALIZE3 / EXP_MAP_TO_QUAT / AA_TO_QUAT axis / ROTATE_2D vector in
 * the first two components); b: second operand ([n][4], or [n][3] for QUAT_ROTATE) or NULL; t: [n] scalar (slerp
 * parameter / angle) or NULL; out: [n][PARC_QOP out width: 4, 3, 6 (tan-norm), 1 or 2]. */
enum {
    PARC_QOP_MUL = 0, PARC_QOP_ROTATE = 1, PARC_QOP_CONJ = 2, PARC_QOP_POS = 3, PARC_QOP_NORMALIZE3 = 4, PARC_QOP_TO_AXIS_ANGLE = 5 /* out [n][4]: axis, angle */,
    PARC_QOP_AA_TO_QUAT = 6, PARC_QOP_EXP_MAP_TO_QUAT = 7, PARC_QOP_TO_EXP_MAP = 8, PARC_QOP_DIFF_ANGLE = 9, PARC_QOP_NORMALIZE = 10,
    PARC_QOP_TO_TAN_NORM = 11, PARC_QOP_SLERP = 12, PARC_QOP_HEADING = 13, PARC_QOP_HEADING_QUAT_INV = 14, PARC_QOP_DIFF = 15,
    PARC_QOP_ROTATE_2D = 16, PARC_QOP_SLERP_RR = 17 /* slerp as k_env_post evaluates it: reduced-range sin / acos polynomials, DESIGN.md 4 */
};
int parc_test_quat_op(int32_t op, const float *a_dev, const float *b_dev, const float *t_dev, int32_t n, float *out_dev, void *stream);

/* The compiler flags the library was built with (set by the build recipe): the loader refuses a library whose flags lack
 * -fno-slp-vectorize / -ffp-contract=off (toolchain note). */
const char *parc_build_flags(void);

/* copy the derived frame tables back (tests): [F][3],[F][3],[F][D] */
int parc_env_get_frame_vel_tables(ParcEnv *env, float *root_vel_host, float *root_ang_vel_host, float *dof_vel_host);

/* Timing of the dominant kernel with hipEvents on the launch stream (bench.py roofline leg). */
int parc_env_profile_step(ParcEnv *env, const float *action_dev, void *stream, int32_t iters, float *avg_ms_out,
                          float *avg_post_kernel_ms_out);

/* The waves of a k_dynamics_wave block hand records to each other through LDS flags; a wait is bounded so that a protocol
 * error cannot hang the GPU.  Number of waits that ever hit the bound on this device (synchronises; must be 0; < 0 = error). */
/* What this handle resolved to, "key=value;..." (dynamics kernel, envs per block, collision points / segments, contact parameters, manifold
 * period and margins, curriculum path, the dev_options it was created with): bench.py records it with every measurement.  The string
 * lives as long as the handle. */
const char *parc_env_describe(ParcEnv *env);

int parc_env_dynamics_timeouts(ParcEnv *env);
/* The same two counters {flag-wait timeouts, manifold drops} as two words of host-mapped pinned memory that the first launch after every
 * step refreshes: readable WITHOUT a synchronisation (stale by at most one step).  NULL when pinned memory could not be mapped. */
const unsigned int *parc_env_health_words(ParcEnv *env);
/* Contact planes the dynamics kernel had no room for since the library was loaded (its per-lane plane list and overflow area were full):
 * must stay 0.  Synchronises the device. */
int parc_env_dynamics_manifold_drops(ParcEnv *env);

/* average duration of k_dynamics in the last parc_env_profile_step call (0 when dynamics is off) */
float parc_env_last_dynamics_ms(ParcEnv *env);

/* `env._episode_length = x` (dm_motion_recorder.py:55 raises it to 1000 s so that only the clip end finishes an episode) */
int parc_env_set_episode_length(ParcEnv *env, float seconds);

/* One control step INCLUDING the reset of the envs it finished (== parc_env_step + parc_env_reset_done) as a single
 * hipGraph launch: the ~10 dependent kernel launches are captured once and replayed, which removes the host-side launch
 * cost that dominates below ~10 000 envs.  The action is read from the buffer bound with parc_env_bind_action (write the
 * policy output there); the graph is re-captured automatically when a setter changes something it bakes in.  Kernel
 * timing events and the recorder are not part of the graph: use parc_env_step for those. */
int parc_env_bind_action(ParcEnv *env, const float *action_dev);
int parc_env_step_reset_graph(ParcEnv *env, void *stream);

/* TD(lambda) returns of a rollout (rl_util.py:7-30; called from ppo_agent._build_train_data): one thread per env walks the
 * T steps backwards,  ret[T-1] = r + g*nv,  ret[i] = r[i] + g*((1 - l_i)*nv[i] + l_i*ret[i+1]),  l_i = lambda*(1 - [done[i] != 0]).
 * All arrays are device pointers laid out [T][N] (the experience buffer's layout); same fp32 operation order as the
 * reference's Python loop, so results are bit-identical to it.  Needs no env handle. */
int parc_td_lambda_return(const float *reward, const float *next_vals, const int32_t *done, float discount, float td_lambda,
                          int32_t T, int32_t N, float *ret_out, void *stream);

/* Observation normalisation fused with the experience-buffer write (normalizer.py:87-90 +
 * experience_buffer.record): one pass reads x [n][dim] and writes  norm = clamp((x - mean) / std, -clip, clip)  and, when
 * copy_out is not NULL, the unmodified row into the rollout buffer slot.  Same fp32 operations in the same order as the
 * three torch kernels it replaces (IEEE division), so `norm` is bit-identical to Normalizer.normalize. */
int parc_normalize_record(const float *x, const float *mean, const float *std, float clip, float *norm_out, float *copy_out,
                          int64_t n, int32_t dim, void *stream);

/* Minibatch gather of the experience buffer (experience_buffer.py:81-89: `{k: v[idx] for k, v in flat_buffers}`, one indexing kernel per
 * buffer in the reference).  ONE launch copies, for every sampled row r < n, row (idx[r] mod count) of each of the
 * num_buffers flat buffers into row r of its contiguous minibatch tensor: src[b] / dst[b] are device pointers, row_bytes[b] the size of one
 * row of buffer b in bytes (any element type: the copy is byte-exact, i.e. bit-identical to index_select).  idx: device int64.  Rows of
 * at least 64 bytes are moved by a wavefront each (16-byte lanes where the row is 16-byte aligned), narrower ones by one thread per row. */
#define PARC_MAX_GATHER_BUFFERS 16
int parc_gather_rows(int32_t num_buffers, const void *const *src_dev, void *const *dst_dev, const int64_t *row_bytes, const int64_t *idx_dev,
                     int64_t n, int64_t count, void *stream);

/* Recorder (IGParkourEnv.write_agent_states, ig_parkour_env.py:759-796; driven by dm_motion_recorder.py:52-121).
 * The reference appends one row per recording env to Python lists every step; here the rows go to device ring buffers
 * owned by the caller:
 *   frames  [cap][N][PARC_REC_WIDTH] f32: root_pos 3 (env-local) | root_rot 4 (xyzw) | joint_rot (B-1) x 4 | contacts B
 *           (contact = |F_b| > 1e-5, _get_char_state ig_parkour_env.py:664-685; the reference stores exp maps / dofs and
 *           converts on save, the motion-terrain file format holds quaternions: file_io.py:8-16)
 *   obs     [cap][N][obs_dim] f32 or NULL (record_obs)
 *   count   [N] i32 rows written; writing [N] u8 (1 = this env is recording); n_writing [1] i32 number of such envs
 * parc_env_record_frame appends the CURRENT state of every recording env (call it after reset for frame 0 and after
 * every step, as the reference does) and ends the recording of envs whose done flag is FAIL; record_ref != 0 stores the
 * reference-motion state instead of the character's (needs the ref_* mirrors of ParcEnvBuffers). */
#define PARC_REC_WIDTH(B) (3 + 4 + 4 * ((B) - 1) + (B))
int parc_env_record_bind(ParcEnv *env, float *frames_dev, float *obs_dev, int32_t cap, int32_t *count_dev, uint8_t *writing_dev,
                         int32_t *n_writing_dev, int32_t record_ref);
int parc_env_record_frame(ParcEnv *env, void *stream);

/* Kernel timing over a run of ordinary parc_env_step calls (what bench.py reports as the roofline's kernel duration).
 * While enabled, every step records three events on the caller's stream (before the dynamics kernel, after it, after
 * the observation kernels); nothing synchronises.  parc_env_get_kernel_timing waits for the last step, returns the
 * average duration of the dynamics kernel and of the observation kernels (k_env_post<STEP>, preceded by k_env_prep when
 * the dynamics kernel did not write the prep records itself) over the steps since the last call, and
 * clears the record. */
int parc_env_set_kernel_timing(ParcEnv *env, int32_t enable);
int parc_env_get_kernel_timing(ParcEnv *env, double *dynamics_ms_avg, double *obs_ms_avg, int32_t *steps);
/* Per-step samples of the same events (call before parc_env_get_kernel_timing, which clears them): dynamics kernel, observation kernel and
 * the curriculum launches of up to `cap` recorded steps; *steps = steps recorded. */
int parc_env_get_kernel_timing_samples(ParcEnv *env, float *dynamics_ms, float *obs_ms, float *curriculum_ms, int32_t cap, int32_t *steps);

/* name of the dynamics kernel this handle launches ("k_dynamics_wave", "k_dynamics_coop", "k_dynamics"; "" when
 * dynamics is off).  The choice follows the shape of the kinematic tree (see parc_env_create). */
const char *parc_env_dynamics_kernel(ParcEnv *env);

/* instantiation of the observation kernel the bound buffers select: "k_env_post<MODE,true>" when any optional output (the
 * ref_* mirrors, ray_hfs, tracking_error) is bound, "k_env_post<MODE,false>" otherwise (the training / bench configuration);
 * "" before parc_env_bind_buffers.  The parity tests assert which one they exercised. */
const char *parc_env_post_kernel(ParcEnv *env);

#ifdef __cplusplus
}
#endif
#endif /* PARC_ENV_H */
